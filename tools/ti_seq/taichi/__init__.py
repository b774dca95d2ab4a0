"""A SEQUENTIAL stand-in for the subset of Taichi that /root/reference/taichi_slam/mapping/{mapping_common,dense_tsdf}.py use, so that the
reference's own source can be imported and RUN in this container (Taichi itself is not installable here: no network) to produce golden
vectors for the oracle -- tools/gen_ref_golden.py.  Development tool: it never travels into tests/, oracle/ or the product.

What it is: `@ti.kernel` / `@ti.func` bodies are executed as the Python they are, one loop iteration after the other, on typed values:
  * scalars are `TV(value, dtype)`; every operation rounds to its result type (numpy scalar arithmetic: an f16 op is computed in f32 and
    rounded once, which is the correctly rounded f16 result; f32 ops are IEEE f32); result types follow Taichi's promotion rules (float
    beats int, wider beats narrower), NOT numpy's; Python literals and Python-scope values are default-typed constants (f32 / i32) and an
    expression of Python values only is evaluated by Python, i.e. at "compile time", exactly as Taichi's front end does; `/` is true
    division (integers are cast to f32 first), `range(x)` truncates x to i32, `ti.round` rounds half away from zero;
  * a local variable keeps the type of its first assignment (later values are cast to it) -- the kernel's source is rewritten with `ast`
    so that assignments go through `__ti_local__`, `field[...] op= v` through `__ti_aug__` (the value is cast to the field's type BEFORE
    the add, as Taichi's atomic ops do) and nothing else is touched;
  * vectors / matrices are lists of TVs with Taichi's evaluation order (`dot`, `norm_sqr`, `@`: products summed left to right, no FMA);
  * fields are numpy arrays; fields placed under `ti.root.pointer(...).dense(...)` live in a dict of dense blocks: a read of an inactive
    cell yields 0 and activates nothing, a write activates the block, a struct-for visits the active blocks in lexicographic order of
    their coordinates and the cells of a block in row-major order (last axis fastest) -- ONE legal serialisation of Taichi's parallel
    struct-for, and the one oracle FAITHFUL replays.
What it is not: Taichi.  Nothing here is parallel, there are no atomics, no fast-math, no FMA contraction; where Taichi's behaviour is a
property of its back end (f16 arithmetic through f32, ti.round, the cast in front of an atomic add) the rule above is this file's reading of
the Taichi documentation, the same reading the oracle states in its own header."""
import ast
import inspect
import math
import textwrap
import types as _pytypes

import numpy as np

np.seterr(all="ignore")


# ---------------------------------------------------------------------------------------------------------------- dtypes
class DType:
    def __init__(self, name, npt, is_float, bits, signed=True):
        self.name, self.np, self.is_float, self.bits, self.signed = name, npt, is_float, bits, signed

    def __repr__(self):
        return f"ti.{self.name}"

    def __call__(self, x):                      # ti.f32(x): a cast
        return cast(x, self)


f16 = DType("f16", np.float16, True, 16)
f32 = DType("f32", np.float32, True, 32)
f64 = DType("f64", np.float64, True, 64)
i8 = DType("i8", np.int8, False, 8)
i16 = DType("i16", np.int16, False, 16)
i32 = DType("i32", np.int32, False, 32)
i64 = DType("i64", np.int64, False, 64)
u8 = DType("u8", np.uint8, False, 8, False)
u16 = DType("u16", np.uint16, False, 16, False)
u32 = DType("u32", np.uint32, False, 32, False)
u64 = DType("u64", np.uint64, False, 64, False)
float16, float32, float64, int8, int16, int32, int64, uint8, uint16, uint32, uint64 = f16, f32, f64, i8, i16, i32, i64, u8, u16, u32, u64
_BY_NP = {np.dtype(d.np): d for d in (f16, f32, f64, i8, i16, i32, i64, u8, u16, u32, u64)}


def _dt(x):
    if isinstance(x, DType):
        return x
    if x is float:
        return f32
    if x is int:
        return i32
    return _BY_NP[np.dtype(x)]


def promote(a, b):
    if a is b:
        return a
    if a.is_float and b.is_float:
        return a if a.bits >= b.bits else b
    if a.is_float:
        return a
    if b.is_float:
        return b
    if a.bits != b.bits:
        return a if a.bits > b.bits else b
    return a if not a.signed else b


# ---------------------------------------------------------------------------------------------------------------- scalars
def _conv(v, dt):
    """value -> numpy scalar of dt with C-like conversion (float -> int truncates)"""
    if dt.is_float:
        return dt.np(v)
    if isinstance(v, (float, np.floating)):
        if not math.isfinite(float(v)):
            return dt.np(0)
        v = int(float(v))                       # truncation towards zero
    v = int(v)
    m = 1 << dt.bits
    v &= m - 1
    if dt.signed and v >= m >> 1:
        v -= m
    return dt.np(v)


class TV:
    __slots__ = ("v", "dt")

    def __init__(self, v, dt):
        self.v, self.dt = _conv(v, dt), dt

    # -- conversions
    def __repr__(self):
        return f"TV({self.v!r}:{self.dt.name})"

    def __int__(self):
        return int(self.v)

    def __index__(self):
        return int(self.v)

    def __float__(self):
        return float(self.v)

    def __bool__(self):
        return bool(self.v)

    def __hash__(self):
        return hash(self.v.item())

    # -- arithmetic
    def _bin(self, o, op, rev=False):
        o = as_tv(o)
        if o is NotImplemented:
            return NotImplemented
        a, b = (o, self) if rev else (self, o)
        dt = promote(a.dt, b.dt)
        if op == "truediv" and not dt.is_float:
            dt = f32                             # integer / integer: both are cast to the default float type
        x, y = _conv(a.v, dt), _conv(b.v, dt)
        if op == "add":
            r = x + y
        elif op == "sub":
            r = x - y
        elif op == "mul":
            r = x * y
        elif op == "truediv":
            r = x / y
        elif op == "floordiv":
            r = np.floor(x / y) if dt.is_float else (x // y if y != 0 else dt.np(0))
        elif op == "mod":
            r = np.fmod(x, y) if dt.is_float else (x % y if y != 0 else dt.np(0))
        elif op == "pow":
            r = x ** y
        else:
            raise NotImplementedError(op)
        return TV(r, dt)

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", True)
    def __truediv__(self, o): return self._bin(o, "truediv")
    def __rtruediv__(self, o): return self._bin(o, "truediv", True)
    def __floordiv__(self, o): return self._bin(o, "floordiv")
    def __rfloordiv__(self, o): return self._bin(o, "floordiv", True)
    def __mod__(self, o): return self._bin(o, "mod")
    def __rmod__(self, o): return self._bin(o, "mod", True)
    def __pow__(self, o): return self._bin(o, "pow")
    def _bit(self, o, f, rev=False):
        o = as_tv(o)
        if o is NotImplemented:
            return NotImplemented
        a, b = (o, self) if rev else (self, o)
        dt = promote(a.dt, b.dt)
        assert not dt.is_float, "bit operation on a float"
        return TV(f(int(a.v), int(b.v)), dt)

    def __or__(self, o): return self._bit(o, lambda a, b: a | b)
    def __ror__(self, o): return self._bit(o, lambda a, b: a | b, True)
    def __and__(self, o): return self._bit(o, lambda a, b: a & b)
    def __rand__(self, o): return self._bit(o, lambda a, b: a & b, True)
    def __xor__(self, o): return self._bit(o, lambda a, b: a ^ b)
    def __rxor__(self, o): return self._bit(o, lambda a, b: a ^ b, True)
    def __lshift__(self, o): return self._bit(o, lambda a, b: a << b)
    def __rlshift__(self, o): return self._bit(o, lambda a, b: a << b, True)
    def __rshift__(self, o): return self._bit(o, lambda a, b: a >> b)
    def __rrshift__(self, o): return self._bit(o, lambda a, b: a >> b, True)
    def __invert__(self): return TV(~int(self.v), self.dt)
    def __neg__(self): return TV(-self.v, self.dt)
    def __pos__(self): return self
    def __abs__(self): return TV(abs(self.v), self.dt)

    # -- comparisons (in the promoted type) -> Python bool
    def _cmp(self, o, f):
        o = as_tv(o)
        if o is NotImplemented:
            return NotImplemented
        dt = promote(self.dt, o.dt)
        return bool(f(_conv(self.v, dt), _conv(o.v, dt)))

    def __lt__(self, o): return self._cmp(o, lambda a, b: a < b)
    def __le__(self, o): return self._cmp(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._cmp(o, lambda a, b: a > b)
    def __ge__(self, o): return self._cmp(o, lambda a, b: a >= b)
    def __eq__(self, o): return self._cmp(o, lambda a, b: a == b)
    def __ne__(self, o): return self._cmp(o, lambda a, b: a != b)


def as_tv(x):
    """a scalar operand as TV: Python / numpy scalars are constants of the default types"""
    if isinstance(x, TV):
        return x
    if isinstance(x, (bool, np.bool_)):
        return TV(int(x), i32)
    if isinstance(x, (int, np.integer)):
        return TV(int(x), i32)
    if isinstance(x, (float, np.floating)):
        return TV(float(x), f32)
    return NotImplemented


def cast(x, dt):
    dt = _dt(dt)
    if isinstance(x, Vec):
        return x.cast(dt)
    if isinstance(x, (list, tuple)):
        return Vec([cast(e, dt) for e in x])
    t = as_tv(x)
    if t is NotImplemented:
        raise TypeError(f"cannot cast {type(x)}")
    return TV(t.v, dt)


# ---------------------------------------------------------------------------------------------------------------- vectors / matrices
def _ew(a, b, f):
    """elementwise over (nested) vectors with scalar broadcast"""
    av, bv = isinstance(a, (Vec, list, tuple)), isinstance(b, (Vec, list, tuple))
    if av and bv:
        assert len(a) == len(b), "vector length mismatch"
        return Vec([_ew(x, y, f) for x, y in zip(a, b)])
    if av:
        return Vec([_ew(x, b, f) for x in a])
    if bv:
        return Vec([_ew(a, y, f) for y in b])
    a, b = as_tv(a), as_tv(b)
    return f(a, b)


class Vec:
    """ti.Vector / ti.Matrix value: a list of TVs (a matrix: a list of row Vecs)"""

    def __init__(self, e):
        self.e = [x if isinstance(x, (TV, Vec)) else (Vec(x) if isinstance(x, (list, tuple)) else as_tv(x)) for x in e]

    def __repr__(self): return f"Vec({self.e})"
    def __len__(self): return len(self.e)
    def __iter__(self): return iter(self.e)

    def __getitem__(self, i):
        if isinstance(i, tuple):
            if isinstance(i[1], slice):                # m[r, :]: a 1 x n matrix (the reference indexes the result as p[0, c])
                return Vec([Vec(list(self.e[int(i[0])].e[i[1]]))])
            return self.e[int(i[0])][int(i[1])]
        return self.e[int(i)]

    def __setitem__(self, i, v):
        if isinstance(i, tuple):
            r = self.e[int(i[0])]
            r[int(i[1])] = v
            return
        old = self.e[int(i)]
        self.e[int(i)] = cast(v, old.dt) if isinstance(old, TV) else v

    @property
    def n(self): return len(self.e)

    def cast(self, dt): return Vec([cast(x, dt) for x in self.e])
    def to_list(self): return [x.to_list() if isinstance(x, Vec) else x for x in self.e]

    def __add__(self, o): return _ew(self, o, lambda a, b: a + b)
    def __radd__(self, o): return _ew(o, self, lambda a, b: a + b)
    def __sub__(self, o): return _ew(self, o, lambda a, b: a - b)
    def __rsub__(self, o): return _ew(o, self, lambda a, b: a - b)
    def __mul__(self, o): return _ew(self, o, lambda a, b: a * b)
    def __rmul__(self, o): return _ew(o, self, lambda a, b: a * b)
    def __truediv__(self, o): return _ew(self, o, lambda a, b: a / b)
    def __rtruediv__(self, o): return _ew(o, self, lambda a, b: a / b)
    def __floordiv__(self, o): return _ew(self, o, lambda a, b: a // b)
    def __mod__(self, o): return _ew(self, o, lambda a, b: a % b)
    def __neg__(self): return Vec([-x for x in self.e])

    def __matmul__(self, o):
        o = o if isinstance(o, Vec) else Vec(o)
        if isinstance(o.e[0], Vec):              # matrix @ matrix
            cols = len(o.e[0])
            return Vec([Vec([_sum([row[k] * o.e[k][c] for k in range(len(o.e))]) for c in range(cols)]) for row in self.e])
        return Vec([_sum([row[k] * o.e[k] for k in range(len(o.e))]) for row in self.e])       # matrix @ vector

    def dot(self, o): return _sum([a * b for a, b in zip(self.e, o)])
    def norm_sqr(self): return _sum([a * a for a in self.e])
    def norm(self, eps=0): return sqrt(self.norm_sqr() + eps) if eps else sqrt(self.norm_sqr())
    def sum(self): return _sum(list(self.e))
    def normalized(self, eps=0):                    # taichi/lang/matrix.py: invlen = 1 / (self.norm() + eps); return invlen * self
        invlen = 1 / (self.norm() + eps)
        return invlen * self

    def cross(self, o):
        a, b = self.e, list(o)
        return Vec([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])

    def transpose(self):
        if not isinstance(self.e[0], Vec):            # a vector is a column: its transpose is a 1 x n matrix
            return Vec([Vec(list(self.e))])
        return Vec([Vec([self.e[r][c] for r in range(len(self.e))]) for c in range(len(self.e[0]))])


def _sum(xs):
    acc = xs[0]
    for x in xs[1:]:
        acc = acc + x
    return acc


class _VectorNS:
    """ti.Vector / ti.Matrix"""

    def __init__(self, matrix):
        self.matrix = matrix

    def __call__(self, e, dt=None):
        v = Vec(list(e))
        return v.cast(_dt(dt)) if dt is not None else v

    def field(self, *a, **kw):
        if self.matrix:
            n, m = a[0], a[1]
            return MatrixField((n, m), _dt(kw.get("dtype", a[2] if len(a) > 2 else f32)), kw.get("shape", a[3] if len(a) > 3 else None))
        n = a[0]
        return MatrixField((n,), _dt(kw.get("dtype", a[1] if len(a) > 1 else f32)), kw.get("shape", a[2] if len(a) > 2 else None))

    def identity(self, dt, n): return Vec([[TV(1 if r == c else 0, _dt(dt)) for c in range(n)] for r in range(n)])

    def zero(self, dt, n, m=None):
        return Vec([TV(0, _dt(dt)) for _ in range(n)]) if m is None else Vec([[TV(0, _dt(dt)) for _ in range(m)] for _ in range(n)])

    def one(self, dt, n): return Vec([TV(1, _dt(dt)) for _ in range(n)])


Vector, Matrix = _VectorNS(False), _VectorNS(True)


# ---------------------------------------------------------------------------------------------------------------- math
def _un(x, f, keep=True):
    if isinstance(x, (Vec, list, tuple)):
        return Vec([_un(e, f, keep) for e in x])
    t = as_tv(x)
    dt = t.dt if t.dt.is_float else f32
    return TV(f(_conv(t.v, dt)), dt)


def sqrt(x): return _un(x, np.sqrt)
def sin(x): return _un(x, np.sin)
def cos(x): return _un(x, np.cos)
def exp(x): return _un(x, np.exp)
def log(x): return _un(x, np.log)
def tanh(x): return _un(x, np.tanh)


def abs(x):                                       # noqa: A001  (ti.abs)
    if isinstance(x, (Vec, list, tuple)):
        return Vec([abs(e) for e in x])
    t = as_tv(x)
    return TV(np.abs(t.v), t.dt)


def _round_half_away(v):
    r = np.trunc(v)
    return r + np.copysign(v.dtype.type(1), v) if np.abs(v - r) >= 0.5 else r


def round(x, dt=None):                            # noqa: A001  (ti.round: half away from zero, C roundf)
    if isinstance(x, (Vec, list, tuple)):
        return Vec([round(e, dt) for e in x])
    t = as_tv(x)
    r = TV(_round_half_away(t.v), t.dt) if t.dt.is_float else t
    return cast(r, dt) if dt is not None else r


def floor(x, dt=None):
    if isinstance(x, (Vec, list, tuple)):
        return Vec([floor(e, dt) for e in x])
    t = as_tv(x)
    r = TV(np.floor(t.v), t.dt) if t.dt.is_float else t
    return cast(r, dt) if dt is not None else r


def ceil(x, dt=None):
    if isinstance(x, (Vec, list, tuple)):
        return Vec([ceil(e, dt) for e in x])
    t = as_tv(x)
    r = TV(np.ceil(t.v), t.dt) if t.dt.is_float else t
    return cast(r, dt) if dt is not None else r


def _minmax(a, b, pick_a):
    if isinstance(a, (Vec, list, tuple)) or isinstance(b, (Vec, list, tuple)):
        return _ew(a, b, lambda x, y: _minmax(x, y, pick_a))
    a, b = as_tv(a), as_tv(b)
    dt = promote(a.dt, b.dt)
    x, y = _conv(a.v, dt), _conv(b.v, dt)
    return TV(x if pick_a(x, y) else y, dt)


def min(*a):                                      # noqa: A001
    r = a[0]
    for x in a[1:]:
        r = _minmax(r, x, lambda p, q: p < q or q != q)
    return r


def max(*a):                                      # noqa: A001
    r = a[0]
    for x in a[1:]:
        r = _minmax(r, x, lambda p, q: p > q or q != q)
    return r


def static(*a):
    return a[0] if len(a) == 1 else a


def random(dtype=float):
    raise NotImplementedError("ti.random is not part of the paths this stand-in runs")


def atomic_add(x, v):
    raise NotImplementedError("ti.atomic_add on an expression: rewritten by the kernel transformer")


def is_active(node, idx):
    return node._is_active(_key(idx))


def loop_config(**kw):
    return None


def grouped(x):
    return ((Vec(list(i)) if not isinstance(i, Vec) else i) for i in x)


def ndrange(*a):
    import itertools
    rs = [__ti_range__(*x) if isinstance(x, tuple) else __ti_range__(x) for x in a]
    return itertools.product(*[list(r) for r in rs])


def init(*a, **kw):
    return None


def sync():
    return None


cpu, gpu, cuda, vulkan = "cpu", "gpu", "cuda", "vulkan"


# ---------------------------------------------------------------------------------------------------------------- fields and SNodes
i, j, k, l = [0], [1], [2], [3]                   # noqa: E741  (axis tokens)
ij, ijk, ijkl, jk, jkl, kl = [0, 1], [0, 1, 2], [0, 1, 2, 3], [1, 2], [1, 2, 3], [2, 3]


def _key(idx):
    """an index expression -> tuple of Python ints"""
    if idx is None:
        return ()
    if isinstance(idx, (TV, int, np.integer)):
        return (int(idx),)
    out = []
    for x in (idx.e if isinstance(idx, Vec) else idx):
        if isinstance(x, (Vec, list, tuple)):
            out.extend(_key(x))
        else:
            out.append(int(x))
    return tuple(out)


class SNode:
    def __init__(self, parent=None, kind="root", axes=(), dims=()):
        self.parent_, self.kind, self.axes, self.dims = parent, kind, list(axes), list(dims)
        self.children, self.fields, self.offset = [], [], None
        self.blocks = {}                          # dense leaf: block coordinates -> {field id: ndarray of the block}

    def _child(self, kind, axes, dims):
        dims = [dims] * len(axes) if isinstance(dims, (int, np.integer)) else list(dims)
        c = SNode(self, kind, axes, dims)
        self.children.append(c)
        return c

    def pointer(self, axes, dims): return self._child("pointer", axes, dims)
    def dense(self, axes, dims): return self._child("dense", axes, dims)
    def bitmasked(self, axes, dims): return self._child("dense", axes, dims)
    def parent(self, n=1): return self.parent_ if n == 1 else self.parent_.parent(n - 1)

    def place(self, *fields, offset=None):
        # (a field placed under a pointer node, as the reference's Octomap does: the node's own cells are the storage blocks)
        chain, n = [], self
        while n is not None and n.kind != "root":
            chain.append(n)
            n = n.parent_
        naxes = 1 + builtins_max(a for c in chain for a in c.axes)
        shape = [1] * naxes
        for c in chain:
            for a, d in zip(c.axes, c.dims):
                shape[a] *= int(d)
        blk = [1] * naxes
        for a, d in zip(self.axes, self.dims):
            blk[a] = int(d)
        if not self.fields:
            self.shape, self.blk = tuple(shape), tuple(blk)
            self.offset = tuple(int(o) for o in offset) if offset is not None else (0,) * naxes
        for f in fields:
            f._place(self)
            self.fields.append(f)

    # -- storage
    def _split(self, key):
        p = tuple(kk - o for kk, o in zip(key, self.offset))
        for a, (x, s) in enumerate(zip(p, self.shape)):
            if x < 0 or x >= s:
                raise IndexError(f"index {key} outside the field (axis {a}: {x} not in [0, {s})) -- undefined behaviour in Taichi; choose inputs that stay inside")
        return tuple(x // b for x, b in zip(p, self.blk)), tuple(x % b for x, b in zip(p, self.blk))

    def _is_active(self, key):
        return self._split(key)[0] in self.blocks

    def _read(self, f, key):
        b, c = self._split(key)
        blk = self.blocks.get(b)
        return None if blk is None else blk[id(f)][c]

    def _write(self, f, key, val):
        b, c = self._split(key)
        blk = self.blocks.get(b)
        if blk is None:
            blk = self.blocks[b] = {id(g): np.zeros(self.blk + g._eshape, g.dtype.np) for g in self.fields}
        blk[id(f)][c] = val

    def _cells(self):
        """active cells in struct-for order: blocks by coordinates, cells row-major"""
        out = []
        for b in sorted(self.blocks):
            base = [bb * s + o for bb, s, o in zip(b, self.blk, self.offset)]
            for c in np.ndindex(*self.blk):
                out.append(tuple(x + y for x, y in zip(base, c)))
        return out

    def __iter__(self):                            # struct-for over the node the fields are placed in: its cells
        return iter(self._cells())

    def deactivate_all(self):
        self.blocks.clear()
        for c in self.children:
            c.deactivate_all()


def builtins_max(it):
    import builtins
    return builtins.max(it)


root = SNode()


class _FieldBase:
    def __init__(self, dtype, shape, eshape=()):
        self.dtype, self._eshape, self.snode, self.arr = dtype, tuple(eshape), None, None
        if shape is not None:
            self.shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
            self.arr = np.zeros(self.shape + self._eshape, dtype.np)

    def _place(self, node):
        self.snode = node
        self.shape = node.shape

    def parent(self, n=1):
        # n = 0: the field's own place node, n = 1: the node it is placed in -- both walk the field's cells.  (A struct-for over a COARSER level of a
        # pointer tree is not restated here: the reference's callers use level 0, scripts/taichislam_node.py:37.)
        if n in (0, 1):
            return self if n == 0 else self.snode
        raise NotImplementedError("struct-for over an ancestor above the field's own node")

    def _get(self, key):
        if self.arr is not None:
            return self.arr[key]
        r = self.snode._read(self, key)
        return np.zeros(self._eshape, self.dtype.np)[()] if r is None else r

    def _put(self, key, val):
        if self.arr is not None:
            self.arr[key] = val
        else:
            self.snode._write(self, key, val)

    def __iter__(self):                            # struct-for
        if self.arr is not None:
            cells = list(np.ndindex(*self.shape))
        else:
            cells = self.snode._cells()
        for c in cells:
            yield tuple(TV(x, i32) for x in c) if len(c) > 1 else TV(c[0], i32)

    def to_numpy(self):
        assert self.arr is not None, "to_numpy of a sparse field is not supported by the stand-in"
        return self.arr.copy()

    def from_numpy(self, a): self.arr[...] = a
    def fill(self, v): self.arr[...] = v


class ScalarField(_FieldBase):
    def __getitem__(self, idx): return TV(self._get(_key(idx)), self.dtype)
    def __setitem__(self, idx, v): self._put(_key(idx), cast(v, self.dtype).v)


class ElemRef(Vec):
    """the vector / matrix stored at one field index: reads like a Vec, element writes go through to the field"""

    def __init__(self, fld, key):
        self.fld, self.key = fld, key
        a = fld._get(key)
        self.e = [TV(x, fld.dtype) for x in a] if a.ndim == 1 else [Vec([TV(x, fld.dtype) for x in r]) for r in a]

    def __setitem__(self, i, v):
        a = np.array(self.fld._get(self.key))
        ii = tuple(int(x) for x in i) if isinstance(i, tuple) else int(i)
        a[ii] = cast(v, self.fld.dtype).v
        self.fld._put(self.key, a)
        self.e = [TV(x, self.fld.dtype) for x in a] if a.ndim == 1 else [Vec([TV(x, self.fld.dtype) for x in r]) for r in a]


class MatrixField(_FieldBase):
    def __init__(self, eshape, dtype, shape):
        super().__init__(dtype, shape, eshape)

    def __getitem__(self, idx): return ElemRef(self, _key(idx))

    def __setitem__(self, idx, v):
        v = v if isinstance(v, Vec) else Vec(list(v))
        a = np.array([[cast(x, self.dtype).v for x in r] for r in v.e] if isinstance(v.e[0], Vec) else [cast(x, self.dtype).v for x in v.e], self.dtype.np)
        if a.shape != self._eshape and a.size == int(np.prod(self._eshape)):
            a = a.reshape(self._eshape)                # (an n x 1 or 1 x n matrix into a vector field)
        assert a.shape == self._eshape, f"shape {a.shape} into a field of {self._eshape}"
        self._put(_key(idx), a)


def field(dtype, shape=None, **kw):
    return ScalarField(_dt(dtype), shape)


# ---------------------------------------------------------------------------------------------------------------- kernel arguments
class NdArr:
    """ti.types.ndarray() argument: a numpy array seen from kernel scope"""

    def __init__(self, a, element_dim=0):
        self.a, self.element_dim = a, element_dim
        self.shape = tuple(int(s) for s in (a.shape[:a.ndim - element_dim] if element_dim else a.shape))

    def __getitem__(self, idx):
        v = self.a[_key(idx)]
        if isinstance(v, np.ndarray):
            return Vec([TV(x, _dt(v.dtype)) for x in v])
        return TV(v, _dt(self.a.dtype))

    def __setitem__(self, idx, v):
        dt = _dt(self.a.dtype)
        if isinstance(v, (Vec, list, tuple)):
            self.a[_key(idx)] = [cast(x, dt).v for x in v]
        else:
            self.a[_key(idx)] = cast(v, dt).v


class _NdAnno:
    def __init__(self, **kw): self.element_dim = kw.get("element_dim", 0)


class _Template:
    pass


def template(): return _Template()


types = _pytypes.SimpleNamespace(ndarray=lambda **kw: _NdAnno(**kw), vector=lambda n, dt: None, matrix=lambda n, m, dt: None)


# ---------------------------------------------------------------------------------------------------------------- kernel / func
class __ti_range__:                                # noqa: N801
    """range() in kernel scope: bounds are cast to i32 (a float bound is truncated), the loop variable is an i32"""

    def __init__(self, *a):
        v = [int(as_tv(x).v) if as_tv(x).dt.is_float is False else int(float(as_tv(x).v)) for x in a]
        self.r = range(*v)

    def __iter__(self): return (TV(x, i32) for x in self.r)
    def __len__(self): return len(self.r)


def __ti_local__(tab, name, v):
    """assignment to a local: the first one fixes the variable's type, later ones are cast to it"""
    if isinstance(v, (bool, np.bool_)):
        return v
    if isinstance(v, (int, float, np.integer, np.floating)):
        v = as_tv(v)
    if isinstance(v, ElemRef):
        v = Vec(list(v.e))                         # a value, not a reference into the field
    if isinstance(v, TV):
        t = tab.get(name)
        if isinstance(t, DType):
            return TV(v.v, t)
        tab[name] = v.dt
        return v
    if isinstance(v, Vec) and v.e and isinstance(v.e[0], TV):
        t = tab.get(name)
        if isinstance(t, tuple) and len(t) == len(v.e):
            return Vec([TV(x.v, d) for x, d in zip(v.e, t)])
        tab[name] = tuple(x.dt for x in v.e)
        return Vec(list(v.e))
    return v


def __ti_aug__(obj, idx, op, v):
    """field[idx] op= v: the value is cast to the field's type first, then one operation in that type (Taichi's atomic ops)"""
    if isinstance(obj, (ScalarField, NdArr)):
        cur = obj[idx]
        vv = cast(v, cur.dt)
        obj[idx] = {"+": cur + vv, "-": cur - vv, "*": cur * vv, "/": cur / vv}[op]
        return
    if isinstance(obj, MatrixField):
        cur = obj[idx]
        vv = cast(v if isinstance(v, (Vec, list, tuple)) else [v] * len(cur), obj.dtype) if not isinstance(v, (Vec, list, tuple)) else cast(v, obj.dtype)
        obj[idx] = {"+": cur + vv, "-": cur - vv, "*": cur * vv, "/": cur / vv}[op]
        return
    if isinstance(obj, Vec):                        # a local vector's element (or an ElemRef: writes through)
        cur = obj[idx]
        obj[idx] = {"+": cur + v, "-": cur - v, "*": cur * v, "/": cur / v}[op]
        return
    raise TypeError(f"augmented assignment into {type(obj)}")


class _Rewrite(ast.NodeTransformer):
    OPS = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/"}

    def _wrap(self, name, value):
        return ast.Call(ast.Name("__ti_local__", ast.Load()), [ast.Name("__ti_vars__", ast.Load()), ast.Constant(name), value], [])

    def visit_Assign(self, node):
        self.generic_visit(node)
        if len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            node.value = self._wrap(node.targets[0].id, node.value)
            return node
        if len(node.targets) == 1 and isinstance(node.targets[0], ast.Tuple) and all(isinstance(t, ast.Name) for t in node.targets[0].elts):
            names = [t.id for t in node.targets[0].elts]
            tmp = ast.Assign([ast.Name("__ti_tmp__", ast.Store())], node.value)
            outs = [tmp]
            for n_, name in enumerate(names):
                outs.append(ast.Assign([ast.Name(name, ast.Store())], self._wrap(name, ast.Subscript(ast.Name("__ti_tmp__", ast.Load()), ast.Constant(n_), ast.Load()))))
            return outs
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        op = self.OPS.get(type(node.op))
        if isinstance(node.target, ast.Name):            # (any operator: +=, |=, <<= ...)
            val = ast.BinOp(ast.Name(node.target.id, ast.Load()), node.op, node.value)
            return ast.Assign([ast.Name(node.target.id, ast.Store())], self._wrap(node.target.id, val))
        if isinstance(node.target, ast.Subscript) and op:
            tgt = node.target
            return ast.Expr(ast.Call(ast.Name("__ti_aug__", ast.Load()), [tgt.value, tgt.slice, ast.Constant(op), node.value], []))
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        # ti.atomic_add(target, v) as an expression: yields the old value, then adds
        if isinstance(node.func, ast.Attribute) and node.func.attr == "atomic_add" and len(node.args) == 2:
            tgt, v = node.args
            if isinstance(tgt, ast.Name):
                new = self._wrap(tgt.id, ast.BinOp(ast.Name(tgt.id, ast.Load()), ast.Add(), v))
                return ast.Subscript(ast.Tuple([ast.Name(tgt.id, ast.Load()), ast.NamedExpr(ast.Name(tgt.id, ast.Store()), new)], ast.Load()), ast.Constant(0), ast.Load())
            if isinstance(tgt, ast.Subscript):
                return ast.Call(ast.Name("__ti_atomic_add__", ast.Load()), [tgt.value, tgt.slice, v], [])
        return node


def __ti_atomic_add__(obj, idx, v):
    old = obj[idx]
    __ti_aug__(obj, idx, "+", v)
    return old


_SELF_GLOBALS = globals()


def _compile(fn):
    src = textwrap.dedent(inspect.getsource(fn))
    tree = ast.parse(src)
    fdef = tree.body[0]
    fdef.decorator_list = []
    fdef.returns = None
    for a in fdef.args.args + fdef.args.kwonlyargs:
        a.annotation = None
    fdef = _Rewrite().visit(fdef)
    fdef.body.insert(0, ast.Assign([ast.Name("__ti_vars__", ast.Store())], ast.Dict([], [])))
    mod = ast.Module([fdef], [])
    ast.fix_missing_locations(mod)
    g = dict(fn.__globals__)
    g.update({"range": __ti_range__, "__ti_local__": __ti_local__, "__ti_aug__": __ti_aug__, "__ti_atomic_add__": __ti_atomic_add__,
              "int": _int, "float": _float, "abs": _pyabs, "max": _pymax, "min": _pymin})
    code = compile(mod, inspect.getsourcefile(fn) or "<ti_seq>", "exec")
    exec(code, g)
    return g[fn.__name__]


def _int(x): return TV(int(float(x.v)) if isinstance(x, TV) and x.dt.is_float else int(x), i32)
def _float(x): return cast(x, f32)
def _pyabs(x): return abs(x) if isinstance(x, (TV, Vec)) else __builtins_abs(x)
def _pymax(*a): return max(*a) if builtins_any(isinstance(x, (TV, Vec)) for x in a) else __builtins_max(*a)
def _pymin(*a): return min(*a) if builtins_any(isinstance(x, (TV, Vec)) for x in a) else __builtins_min(*a)


import builtins as _b                              # noqa: E402
__builtins_abs, __builtins_max, __builtins_min, builtins_any = _b.abs, _b.max, _b.min, _b.any


class _Callable:
    def __init__(self, fn, is_kernel):
        self.fn, self.is_kernel, self.impl = fn, is_kernel, None
        self.sig = inspect.signature(fn)
        self.__name__ = fn.__name__

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return lambda *a, **kw: self(obj, *a, **kw)

    def __call__(self, *a, **kw):
        if self.impl is None:
            self.impl = _compile(self.fn)
        if self.is_kernel:                           # ndarray arguments are wrapped, scalar arguments are cast to their annotation
            params = list(self.sig.parameters.values())
            a = list(a)
            for n_, (p, v) in enumerate(zip(params, a)):
                an = p.annotation
                if isinstance(an, _NdAnno):
                    a[n_] = NdArr(np.asarray(v), an.element_dim)
                elif isinstance(an, DType):
                    a[n_] = cast(v, an)
            r = self.impl(*a, **kw)
            ra = self.sig.return_annotation
            if isinstance(ra, DType) and r is not None:
                r = cast(r, ra).v.item()
            return r
        return self.impl(*a, **kw)


def kernel(fn): return _Callable(fn, True)
def func(fn): return _Callable(fn, False)
def data_oriented(cls): return cls
def pyfunc(fn): return fn
