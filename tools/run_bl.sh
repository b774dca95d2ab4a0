mkdir -p gpurun_out/r5i
run() { name=$1; shift; env "$@" BL_OUT=r5i/$name BL_REPS=${REPS:-60} timeout 500 python tools/repro_r5.py benchlike > gpurun_out/r5i/$name.log 2> gpurun_out/r5i/$name.err; echo "$name rc=$? ok=$(grep -c '"ok": true' gpurun_out/r5i/$name.log) bad=$(grep -c '"ok": false' gpurun_out/r5i/$name.log)"; grep seq_verify gpurun_out/r5i/$name.err | head -30; }
run f1 X=1
run f2 X=1
run f3 X=1
run f4 TSL_SEQ_VERIFY=1
run f5 X=1
