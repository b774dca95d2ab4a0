"""The sequential Taichi stand-in (tools/ti_seq) that produced tests/golden/ref_*.npz: its own rules, checked on small kernels -- type promotion,
one rounding per operation, locals typed by their first assignment, the cast in front of `field[...] += v`, range() truncation, ti.round, sparse
fields (read of an inactive cell, activation by writing, struct-for order)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ti():
    spec = importlib.util.spec_from_file_location("ti_seq_taichi", os.path.join(ROOT, "tools", "ti_seq", "taichi", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


ti = _ti()
f16, f32 = np.float16, np.float32


def test_promotion_and_one_rounding_per_operation():
    a, b = ti.TV(0.1, ti.f16), ti.TV(0.2, ti.f16)
    s = a + b
    assert s.dt is ti.f16 and s.v == f16(f16(0.1) + f16(0.2))
    p = a * ti.TV(3.0, ti.f32)
    assert p.dt is ti.f32 and p.v == f32(f32(f16(0.1)) * f32(3.0))                 # f16 (op) f32 -> f32, the f16 operand converted exactly
    q = ti.TV(7, ti.i32) / ti.TV(2, ti.i32)
    assert q.dt is ti.f32 and q.v == f32(3.5)                                      # integer / integer: true division in the default float type
    assert (ti.TV(7, ti.i32) // ti.TV(2, ti.i32)).v == 3 and (ti.TV(1, ti.i32) + 1.5).dt is ti.f32
    assert (a * 2).dt is ti.f16 and (a * 2.0).dt is ti.f32                          # literals are constants of the default types (i32 / f32)
    assert (ti.TV(3, ti.i32) < 3.5) and not (ti.TV(4, ti.i32) < 3.5)
    v = ti.Vector([3.0, 4.0, 12.0], ti.f16)
    assert v.norm().dt is ti.f16 and v.norm().v == f16(13.0) and v.dot(v).v == f16(169.0)
    n = ti.Vector([0.3, 0.4, 1.2], ti.f16).normalized()                             # invlen = 1 / norm (f16), then invlen * v
    h = [f16(0.3), f16(0.4), f16(1.2)]
    nrm = f16(np.sqrt(f32(f16(f16(h[0] * h[0]) + f16(h[1] * h[1])) + f16(h[2] * h[2]))))
    assert [x.v for x in n] == [f16(f16(f32(1) / f32(nrm)) * x) for x in h]


def test_round_floor_cast_and_range():
    assert [int(ti.round(ti.TV(x, ti.f32), ti.i32)) for x in (0.5, 1.5, 2.5, -0.5, -1.5, 2.49)] == [1, 2, 3, -1, -2, 2]     # half away from zero
    assert int(ti.floor(ti.TV(-1.25, ti.f32), ti.i32)) == -2 and int(ti.cast(ti.TV(-1.75, ti.f32), ti.i32)) == -1               # casts truncate
    assert [int(i) for i in ti.__ti_range__(ti.TV(2.9, ti.f32))] == [0, 1] and [int(i) for i in ti.__ti_range__(1, 7.99)] == [1, 2, 3, 4, 5, 6]
    assert ti.min(ti.TV(2.0, ti.f16), 1000).dt is ti.f16 and ti.min(ti.TV(1500.0, ti.f32), 1000).v == f32(1000.0)


class _K:
    def __init__(self):
        self.acc = ti.field(ti.f16, shape=4)
        self.sp = ti.field(ti.i32)
        self.node = ti.root.pointer(ti.ij, (3, 3)).dense(ti.ij, (2, 2))
        self.node.place(self.sp, offset=[-3, -3])
        self.out = ti.field(ti.f32, shape=4)

    @ti.kernel
    def typed_locals(self, x: ti.f32):
        r = 0.0                      # an f32 variable
        h = self.acc[0]              # an f16 variable
        h = x                        # ... later values are cast to it
        r = h * 3                    # f16 * i32 -> f16, stored into the f32 variable
        self.out[0] = r
        c = 0                        # an i32 variable: the float is truncated
        c = x
        self.out[1] = c
        for _ in range(x):           # range(f32) truncates
            self.out[2] += 1.0

    @ti.kernel
    def accumulate(self, v: ti.f32):
        self.acc[1] += v             # v is cast to f16 FIRST, then one f16 addition
        self.acc[1] += v

    @ti.kernel
    def sparse(self):
        self.out[3] = self.sp[2, 2]              # inactive: reads 0, activates nothing
        self.sp[2, -3] = 7
        self.sp[-3, 1] = 5
        self.sp[-2, 2] = 6


def test_locals_keep_the_type_of_their_first_assignment():
    k = _K()
    k.typed_locals(2.7001953125)
    assert k.out.to_numpy()[0] == f32(f16(f16(2.7001953125) * f16(3))) and k.out.to_numpy()[1] == 2 and k.out.to_numpy()[2] == 2


def test_augmented_assignment_into_a_field_casts_the_value_first():
    k = _K()
    k.acc[1] = 1.0
    v = f32(0.007323588710278273)             # f16(1 + f16(v)) = 1.008, f16(f32(1) + v) = 1.007: the two readings differ here
    k.accumulate(v)
    once = f16(f16(1.0) + f16(v))
    assert once != f16(f32(1.0) + v)
    assert k.acc.to_numpy()[1] == f16(once + f16(v))


def test_sparse_field_semantics_and_struct_for_order():
    k = _K()
    k.sparse()
    assert k.out.to_numpy()[3] == 0 and len(k.node.blocks) == 2        # (-3, 1) and (-2, 2) share a 2 x 2 block; the read of (2, 2) activated nothing
    cells = [tuple(int(x) for x in c) for c in k.sp]
    # blocks in lexicographic order of their coordinates (x -3..-2, y 1..2 is block (0, 2); (2, -3) lies in block (2, 0)), the cells of a block row-major
    blocks = [(c[0] + 3) // 2 * 3 + (c[1] + 3) // 2 for c in cells]
    assert blocks == sorted(blocks) and len(cells) == 8
    first = cells[:4]
    assert first == [(-3, 1), (-3, 2), (-2, 1), (-2, 2)]
    assert [int(k.sp[c]) for c in cells if int(k.sp[c])] == [5, 6, 7]
    # a struct-for over the field's own node (parent(0)) or the node it is placed in (parent(1)) walks the same cells (the reference's Octomap export,
    # taichi_octomap.py:94 with level 0 / 1); a coarser ancestor is not restated
    assert [tuple(int(x) for x in c) for c in ti.grouped(k.sp.parent(0))] == cells == [tuple(int(x) for x in c) for c in ti.grouped(k.sp.parent(1))]
    import pytest
    with pytest.raises(NotImplementedError):
        k.sp.parent(2)
    k.node.parent().deactivate_all()
    assert list(k.sp) == []
