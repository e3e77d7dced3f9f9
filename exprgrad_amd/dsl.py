"""Python mirror of exprgrad's graph-building front-end, for the tests and the benchmark.

The reference builds programs with a Nim macro DSL (`out[y, x] ++= a[y, it] * b[it, x]`,
exprgrad/parser.nim:677-681, exprgrad/dsl.nim).  That front-end is not part of the hot path and
stays the reference's; what crosses the drop-in boundary is the *kernel description* it
produces (ir.nim:211-230).  This module produces exactly that description — as the text
`eg_model_compile` consumes — from Python statements that read like the reference's:

    y, x, it = iters("y x it")
    c = Fun()
    c[y, x] += input("a")[y, it] * input("b")[it, x]          # c*[y,x] ++= a[y,it] * b[it,x]
    hr = Fun()
    hr.raw[it] += select(h.raw[it] <= 0.0, 0.1 * h.raw[it], h.raw[it])   # hr*{it} ++= ...
    model = compile(c.target("c"), gpu=ctx)

Semantics mirrored (with the reference location):
  * expression builders, typed Scalar / Index / Boolean                    dsl.nim:21-146
  * register numbering = post-order build of the value expression, one register per builder
    node per block (shared sub-expressions such as sq(x) = x * x reuse it)  parser.nim:159-217
  * loops are created in order of first use of an iterator: value first, then write index
                                                                            parser.nim:183-196, 231-242
  * Fun graph node kinds and flattening order (children first, effects, generators)
                                                                            parser.nim:72-97, 317-384
  * backwards / grad / optimize / backprop / target                         parser.nim:738-815
"""
import itertools

SCALAR, INDEX, BOOLEAN = "scalar", "index", "boolean"

# instruction kinds, spelled as in ir.nim:51-76 without the Instr prefix, lower case
_ARITH = {"add", "sub", "mul", "div", "indexdiv", "mod", "wrap", "negate", "sin", "cos", "exp", "pow", "sqrt",
          "log", "log10", "log2", "ln"}


class ParserError(Exception):
    """parser.nim ParserError."""


class Expr:
    """ExprBuilder (parser.nim:27-47): kind in {instr, iter, read}."""

    __slots__ = ("kind", "instr", "children", "typ", "lit", "tensor", "is_raw", "dim", "iter", "bounds", "_res")

    def __init__(self, kind, typ, instr=None, children=(), lit=None, tensor=None, is_raw=False, dim=0, iter=None,
                 bounds=None):
        self.kind, self.typ, self.instr = kind, typ, instr
        self.children = list(children)
        self.lit, self.tensor, self.is_raw, self.dim, self.iter, self.bounds = lit, tensor, is_raw, dim, iter, bounds
        self._res = {}

    # ---- arithmetic (dsl.nim:41-72) -------------------------------------------------------
    def _coerce(self, other):
        if isinstance(other, Expr):
            return other
        if isinstance(other, bool):
            return literal(other)
        if self.typ == INDEX and isinstance(other, int):
            return literal(int(other))
        if self.typ == SCALAR and isinstance(other, (int, float)):
            return literal(float(other))
        raise TypeError(f"cannot combine {self.typ} expression with {other!r}")

    def _bin(self, other, instr, typ=None, swap=False):
        other = self._coerce(other)
        if other.typ != self.typ:
            raise TypeError(f"{instr}: operand types differ ({self.typ} vs {other.typ})")
        a, b = (other, self) if swap else (self, other)
        return Expr("instr", typ or self.typ, instr=instr, children=[a, b])

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", swap=True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", swap=True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", swap=True)

    def __truediv__(self, o):
        if self.typ != SCALAR:
            raise TypeError("`/` is defined for Scalar only (dsl.nim:55); use // for Index")
        return self._bin(o, "div")

    def __rtruediv__(self, o): return self._bin(o, "div", swap=True)
    def __floordiv__(self, o): return self._bin(o, "indexdiv")
    def __mod__(self, o): return self._bin(o, "mod")
    def __neg__(self): return Expr("instr", self.typ, instr="negate", children=[self])

    # comparisons: only ==, <, <= exist (dsl.nim:35, 45-46); a > b is b < a, a >= b is b <= a
    def __lt__(self, o): return self._bin(o, "lt", BOOLEAN)
    def __le__(self, o): return self._bin(o, "le", BOOLEAN)
    def __gt__(self, o): return self._bin(o, "lt", BOOLEAN, swap=True)
    def __ge__(self, o): return self._bin(o, "le", BOOLEAN, swap=True)
    def eq(self, o): return self._bin(o, "eq", BOOLEAN)
    __hash__ = object.__hash__

    def __and__(self, o): return self._bin(o, "and", BOOLEAN)
    def __or__(self, o): return self._bin(o, "or", BOOLEAN)

    def __iadd__(self, value):
        """`tensor[dims] += value` is the `++=` statement: returns a marker Fun.__setitem__ consumes."""
        if self.kind != "read":
            raise ParserError("`+=` on an expression is only valid as `tensor[index] += value`")
        if not isinstance(value, Expr):
            value = literal(float(value))
        return _Accumulate(self, value)

    def __bool__(self):
        raise TypeError("expression builders have no truth value; use select(cond, a, b)")


class _Accumulate:
    def __init__(self, target, value):
        self.target, self.value = target, value


def literal(value):
    """parser.nim:104-121."""
    if isinstance(value, Expr):
        return value
    if isinstance(value, bool):
        return Expr("instr", BOOLEAN, instr="boolean", lit=bool(value))
    if isinstance(value, int):
        return Expr("instr", INDEX, instr="index", lit=int(value))
    return Expr("instr", SCALAR, instr="scalar", lit=float(value))


def _scalar(x):
    return x if isinstance(x, Expr) else literal(float(x))


def _unop(name):
    def f(a):
        a = _scalar(a)
        return Expr("instr", SCALAR, instr=name, children=[a])
    f.__name__ = name
    return f


sin, cos, exp, sqrt, ln, log10, log2 = (_unop(n) for n in ("sin", "cos", "exp", "sqrt", "ln", "log10", "log2"))


def pow(a, b):  # noqa: A001 - mirrors dsl.nim:59
    return Expr("instr", SCALAR, instr="pow", children=[_scalar(a), _scalar(b)])


def log(x, base):
    return Expr("instr", SCALAR, instr="log", children=[_scalar(x), _scalar(base)])


def to_scalar(i):
    return Expr("instr", SCALAR, instr="toscalar", children=[literal(i)])


def to_index(s):
    return Expr("instr", INDEX, instr="toindex", children=[_scalar(s)])


toScalar, toIndex = to_scalar, to_index


def epoch():
    return Expr("instr", INDEX, instr="epoch")


def select(cond, a, b):
    """dsl.nim:79-83."""
    if not isinstance(a, Expr) and not isinstance(b, Expr):
        a, b = literal(float(a)), literal(float(b))
    elif not isinstance(a, Expr):
        a = b._coerce(a)
    elif not isinstance(b, Expr):
        b = a._coerce(b)
    return Expr("instr", a.typ, instr="select", children=[cond, a, b])


def sq(x):
    return x * x  # dsl.nim:135-136: the SAME builder twice -> one register used twice


class ArrayLiteral:
    """`literal([1.0, 2.0, 3.0])` / nested arrays (parser.nim:104-121 with array nodes; the reference
    keeps InstrArray / InstrArrayRead / InstrArrayLen in its IR, ir.nim:60-62).  The kernel-description
    text has no array instructions: `arr[i]` is lowered here to the select chain
    select(i == 0, a0, select(i == 1, a1, ...)) over the same literals, `arr.len` to an index literal —
    identical values for every in-range index."""

    def __init__(self, values):
        self.values = [ArrayLiteral(v) if isinstance(v, (list, tuple)) else v for v in values]
        if not self.values:
            raise ParserError("an array literal needs at least one element")

    def len(self):  # noqa: A003
        return literal(len(self.values))

    def __len__(self):
        return len(self.values)

    def __getitem__(self, index):
        index = literal(index)
        if index.typ != INDEX:
            raise ParserError("array literals are indexed with Index expressions")
        picked = self.values
        if isinstance(picked[0], ArrayLiteral):   # arr[y] of a nested array: a view that waits for the next index
            return _ArrayRow(self, index)
        out = literal(float(picked[-1]))
        for i in range(len(picked) - 2, -1, -1):
            out = select(index.eq(i), literal(float(picked[i])), out)
        return out


class _ArrayRow:
    def __init__(self, array, row_index):
        self.array, self.row_index = array, row_index

    def __getitem__(self, index):
        rows = [row[index] for row in self.array.values]
        out = rows[-1]
        for i in range(len(rows) - 2, -1, -1):
            out = select(self.row_index.eq(i), rows[i], out)
        return out

    def len(self):  # noqa: A003
        return literal(len(self.array.values[0]))


def array(values):
    return ArrayLiteral(values)


def max(x, y):  # noqa: A001 - dsl.nim:138-139
    x, y = _scalar(x), _scalar(y)
    return select(x > y, x, y)


def min(x, y):  # noqa: A001 - dsl.nim:141-142
    x, y = _scalar(x), _scalar(y)
    return select(x < y, x, y)


def iters(names):
    """Iterator literals (parser.nim:123-129).  `iters("y x it")` -> three Index builders."""
    out = [Expr("iter", INDEX, iter=n) for n in names.replace(",", " ").split()]
    return out[0] if len(out) == 1 else out


def iter_in(name, start, stop):
    """`x in start..<stop` (parser.nim:609-627): an iterator with explicit bounds."""
    return Expr("iter", INDEX, iter=name, bounds=(literal(start), literal(stop)))


class _Shape:
    def __init__(self, fun):
        self.fun = fun

    def __getitem__(self, dim):
        return Expr("instr", INDEX, instr="shape", tensor=self.fun, dim=int(dim))

    def __len__(self):
        raise TypeError("use shape_len(fun)")


class _Raw:
    def __init__(self, fun):
        self.fun = fun

    def __getitem__(self, index):
        return Expr("read", SCALAR, tensor=self.fun, is_raw=True, children=[literal(index)])

    def __setitem__(self, index, acc):
        self.fun._add_statement(acc, [literal(index)], True)


_fun_ids = itertools.count(1)


class Fun:
    """Graph node (parser.nim:72-97)."""

    def __init__(self, kind="result", name="", **kw):
        self.kind, self.name = kind, name
        self.children = []
        self.kernels = []       # KernelBuilder list (result / effect)
        self.tensor = 0
        self.targets = set()
        self.locked = False
        self.shape_constr = None
        self.uid = next(_fun_ids)
        self.__dict__.update(kw)

    # a[y, x] / a.raw[i]  (dsl.nim:95-107)
    def __getitem__(self, idx):
        idx = idx if isinstance(idx, tuple) else (idx,)
        return Expr("read", SCALAR, tensor=self, is_raw=False, children=[literal(i) for i in idx])

    def __setitem__(self, idx, acc):
        idx = idx if isinstance(idx, tuple) else (idx,)
        self._add_statement(acc, [literal(i) for i in idx], False)

    @property
    def raw(self):
        return _Raw(self)

    @property
    def shape(self):
        return _Shape(self)

    def len(self):
        return Expr("instr", INDEX, instr="len", tensor=self)

    def _add_statement(self, acc, dims, is_raw):
        if not isinstance(acc, _Accumulate):
            raise ParserError("kernels accumulate: write `t[idx] += value` (the reference's `++=`)")
        tgt = acc.target
        if tgt.tensor is not self or tgt.is_raw != is_raw or len(tgt.children) != len(dims):
            raise ParserError("`+=` target mismatch")
        if self.kind == "gradarg":
            # `grad(t)[idx] ++= value` inside a customGrad block (parser.nim:498, 585)
            if not _custom_grad_stack:
                raise ParserError("grad_of(t)[...] += ... is only valid inside `with fun.custom_grad():`")
            block = _custom_grad_stack[-1]
            block.kernels.append(_KernelBuilder(self, dims, is_raw, acc.value))
            _collect_children(acc.value, block.fun)
            return
        if self.kind not in ("result", "effect"):
            raise ParserError("Unable to add a kernel to a " + self.kind)  # parser.nim:437-438
        if self.locked:
            raise ParserError("tensor is locked")
        self.kernels.append(_KernelBuilder(self, dims, is_raw, acc.value))
        _collect_children(acc.value, self)  # parser.nim:430-441

    def custom_grad(self):
        """`do: customGrad: ...` of the statement added last (parser.nim:498, 585; used by maxpool2,
        dnn.nim:59-71): the gradient kernels written inside the block replace derive for it.

            with result.custom_grad():
                grad_of(images)[n, y, x, c] += select(...grad_of(result)[n, y // 2, x // 2, c]...)
        """
        if not self.kernels:
            raise ParserError("custom_grad() follows the statement it belongs to")
        return _CustomGrad(self)

    def reshape(self, shape):
        return reshape(self, shape)

    # ---- shape constraints (parser.nim:683-697) --------------------------------------------
    def copy_shape(self, src):
        if self.kind != "result":
            raise ParserError("Cannot set shape of " + self.kind)
        self.shape_constr = ("copy", src)
        if src not in self.children:
            self.children.append(src)

    copyShape = copy_shape

    def with_shape(self, *dims):
        if self.kind != "result":
            raise ParserError("Cannot set shape of " + self.kind)
        dims = [literal(d) for d in dims]
        self.shape_constr = ("dims", dims)
        for d in dims:
            _collect_children(d, self)

    withShape = with_shape

    def lock(self):
        self.locked = True

    # ---- graph API (parser.nim:738-815) ----------------------------------------------------
    def target(self, name, compile_target="gpu"):
        return Fun("target", name, children=[self], compile_target=compile_target)

    def backwards(self):
        return backwards(self)

    def params(self, stop=()):
        return params(self, set(stop))

    def optimize(self, *args):
        return optimize(self, *args)

    def backprop(self, optim):
        return backprop(self, optim)

    def grad(self, fun):
        return grad(self, fun)


def _collect_children(expr, fun):
    for c in expr.children:
        _collect_children(c, fun)
    if expr.bounds:
        for b in expr.bounds:
            _collect_children(b, fun)
    t = expr.tensor
    if t is not None and t.kind == "gradarg":
        t = t.children[0]  # the gradient placeholder stands for a tensor of the graph
    if t is not None and t is not fun and t not in fun.children:
        fun.children.append(t)


_custom_grad_stack = []


class _CustomGrad:
    def __init__(self, fun):
        self.fun, self.kernels = fun, []

    def __enter__(self):
        _custom_grad_stack.append(self)
        return self

    def __exit__(self, exc_type, exc, tb):
        _custom_grad_stack.pop()
        if exc_type is None:
            self.fun.kernels[-1].grads = self.kernels
        return False


def grad_of(fun):
    """`grad(fun)` inside a customGrad block (FunGradientArg, parser.nim:70, 142-146, 784): the
    gradient tensor of `fun`, bound when the backward pass is generated."""
    return Fun("gradarg", children=[fun])


def _tensor_id(fun):
    return -fun.children[0].tensor if fun.kind == "gradarg" else fun.tensor


def reshape(fun, shape):
    """reshape (parser.nim:786-793; GenReshape, passes.nim:643-688): a raw copy into a tensor whose
    shape is given, one extent may be -1 (= len / product of the others)."""
    shape = [int(d) for d in shape]
    if sum(1 for d in shape if d < 0) > 1:
        raise ParserError("reshape: at most one extent may be -1")
    r = Fun("reshape", "reshape")
    r.children = [fun]
    r.reshape_dims = shape
    return r


def input(name, shape=()):  # noqa: A001 - parser.nim:724-731
    return Fun("input", name, input_shape=[int(s) for s in shape])


def param(shape, init_range=(-0.1, 0.1), name=""):
    """parser.nim:713-722."""
    return Fun("param", name, param_shape=[int(s) for s in shape], init_range=(float(init_range[0]), float(init_range[1])))


def cache(fun, name=""):
    """cache(param, name) (parser.nim:795-798): persistent zero-initialised state shaped like `fun`
    (adam's moment estimates), wrapped in an effect so kernels can accumulate into it."""
    return Fun("effect", effect=Fun("cache", name, cache=fun))


def rand(fun, lo=0.0, hi=1.0):
    """rand(fun, range) (parser.nim:732-736): a tensor shaped like `fun`, refilled with uniform random
    numbers in [lo, hi) on every call (TensorRandom, ir.nim:233-240)."""
    return Fun("random", children=[fun], random_range=(float(lo), float(hi)))


def backwards(fun):
    return Fun("backwards", children=[fun])


def cond(branches, otherwise=None):
    """cond({target name: fun, ...}, otherwise) (parser.nim:812-817): a node that stands for a
    different sub-graph depending on the target being built — the GAN example feeds its
    discriminator the generator's output in `fit.gen` and an input elsewhere (gan.nim:44)."""
    branches = dict(branches)
    return Fun("cond", cond=branches, cond_else=otherwise)


def params(fun, stop=frozenset()):
    """parser.nim:742-756.  (The reference returns a HashSet; here: discovery order, deterministic.)"""
    out = []
    if not (fun.kind == "target" and fun.name in stop):
        subs = list(fun.children)
        if fun.kind == "cond":                               # parser.nim:747-751
            subs += list(fun.cond.values()) + ([fun.cond_else] if fun.cond_else is not None else [])
        for c in subs:
            for p in params(c, stop):
                if p not in out:
                    out.append(p)
        if fun.kind == "param" and fun not in out:
            out.append(fun)
    return out


def optimize(gradients, a, b=None):
    """optimize(gradients, params, optim) / optimize(gradients, optim)   parser.nim:758-778."""
    if b is None:
        plist, optim = params(gradients), a
    else:
        plist, optim = a, b
    result = Fun("multiple")
    for p in plist:
        effect = Fun("effect", effect=p)
        g = Fun("gradient", children=[gradients, p])
        optim(effect, g)
        result.children.append(effect)
    return result


def backprop(loss, optim):
    return optimize(backwards(loss), optim)  # parser.nim:780-781


def grad(gradients, fun):
    return Fun("gradient", children=[gradients, fun])  # parser.nim:783-784


# ------------------------------------------------------------------------------------------------
# Kernel building (parser.nim:159-254) and program flattening (parser.nim:261-417)

class _KernelBuilder:
    def __init__(self, target, dims, is_raw, value):
        self.target, self.dims, self.is_raw, self.value = target, dims, is_raw, value
        self.grads = None  # custom gradient statements (KernelBuilder.grads, parser.nim:57)


class LinearIndex:
    """ir.nim:120-123 after foldLinearIndices: constant + sum(factor * register); `setup`
    instructions only for shape()/len() terms of explicit loop bounds."""

    def __init__(self, constant=0, factors=None, setup=None):
        self.constant, self.factors, self.setup = constant, dict(factors or {}), list(setup or [])

    def tokens(self):
        regs = sorted(r for r, f in self.factors.items() if f != 0)   # kdLinear: register order, deterministic
        out = ["L", str(self.constant), str(len(regs))]
        for r in regs:
            out += [str(r), str(self.factors[r])]
        return out


class Kernel:
    def __init__(self):
        self.nregs = 0
        self.loops = []    # (iter_reg, name, bounds or None)
        self.setup = []    # instrs evaluated on the host: shape/len terms of explicit bounds
        self.reads = []    # (tensor_id, reg, is_raw, [LinearIndex])
        self.index_instrs = []  # like instrs, Index typed, evaluated inside the loop nest before the
                                # reads: the non-affine parts of tensor indices (`y div 2`)
        self.instrs = []   # (kind, res, [args], extra)
        self.result = 0
        self.write = None  # (tensor_id, reg, is_raw, [LinearIndex])
        self.generator = None  # ("backwards", tid) | ("gradient", of_tid, dest_tid)
        self.custom_grad = None  # [Kernel]: used instead of derive (ir.nim:203-209); tensor id -t = gradient of t

    def alloc(self):
        self.nregs += 1
        return self.nregs


class _Ctx:
    def __init__(self):
        self.kernel = Kernel()
        self.iters = {}
        self.blocks = 0

    def block(self):
        self.blocks += 1
        return self.blocks - 1


def _build(expr, instrs, block, ctx):
    """parser.nim:159-217."""
    if block in expr._res:
        return expr._res[block]
    k = ctx.kernel
    if expr.kind == "read":
        dims = [_build_linear(d, ctx) for d in expr.children]
        res = k.alloc()
        k.reads.append((_tensor_id(expr.tensor), res, expr.is_raw, dims))
    elif expr.kind == "iter":
        if expr.iter not in ctx.iters:
            reg = k.alloc()
            ctx.iters[expr.iter] = reg
            bounds = None
            if expr.bounds:
                bounds = (_build_linear(expr.bounds[0], ctx), _build_linear(expr.bounds[1], ctx))
            k.loops.append((reg, expr.iter, bounds))
        res = ctx.iters[expr.iter]
    else:
        args = [_build(c, instrs, block, ctx) for c in expr.children]
        extra = None
        if expr.instr in ("index", "scalar", "boolean"):
            extra = expr.lit
        elif expr.instr == "shape":
            extra = (expr.tensor.tensor, expr.dim)
        elif expr.instr in ("len", "shapelen"):
            extra = (expr.tensor.tensor,)
        res = k.alloc()
        instrs.append((expr.instr, res, args, extra))
    expr._res[block] = res
    return res


def _lin_scale(a, b):
    """LinearIndex * int (ir.nim:626-631): zero when b == 0."""
    if b == 0:
        return LinearIndex()
    return LinearIndex(a.constant * b, {r: f * b for r, f in a.factors.items()})


def _lin_add(a, b):
    """LinearIndex + LinearIndex (ir.nim:633-643)."""
    out = LinearIndex(a.constant + b.constant, a.factors)
    for r, f in b.factors.items():
        if r in out.factors:
            out.factors[r] += f
            if out.factors[r] == 0:
                del out.factors[r]
        else:
            out.factors[r] = f
    return out


def _fold_setup(setup, reg, ctx):
    """foldSetup (passes.nim:195-244): the instructions an index expression was built into (every node of the
    expression got a register, parser.nim:155-217) collapse into constant + sum(factor * register) as far as they are
    affine in the iterators; what is not (`y div 2`, shape(), len(), a product of two iterators) stays behind as setup
    instructions whose result enters with factor 1.  Register numbers keep the gaps the folded instructions leave."""
    regs = {r: LinearIndex(0, {r: 1}) for r in ctx.iters.values()}
    get = lambda r: regs.get(r, LinearIndex())
    for (kind, res, args, extra) in setup:
        if kind == "index":
            regs[res] = LinearIndex(int(extra))
        elif kind == "add":
            regs[res] = _lin_add(get(args[0]), get(args[1]))
        elif kind == "sub":
            regs[res] = _lin_add(get(args[0]), _lin_scale(get(args[1]), -1))
        elif kind == "negate":
            regs[res] = _lin_scale(get(args[0]), -1)
        elif kind == "mul" and not get(args[0]).factors:
            regs[res] = _lin_scale(get(args[1]), get(args[0]).constant)
        elif kind == "mul" and not get(args[1]).factors:
            regs[res] = _lin_scale(get(args[0]), get(args[1]).constant)
        else:
            regs[res] = LinearIndex(0, {res: 1})
    total = get(reg)
    used = set(total.factors)
    kept = []
    for ins in reversed(setup):
        if ins[1] in used:
            kept.append(ins)
            used.update(ins[2])
    kept.reverse()
    return LinearIndex(total.constant, total.factors, kept)


def _build_linear(expr, ctx, fold=True):
    """buildLinearIndex (parser.nim:155-157) followed by foldSetup.  fold=False: shape constraints, which
    foldLinearIndices never visits (passes.nim:246-262 walks kernels only) — they keep every instruction."""
    if expr.typ != INDEX:
        raise ParserError("tensor indices must be Index expressions")
    setup = []
    reg = _build(expr, setup, ctx.block(), ctx)
    if not fold:
        return LinearIndex(0, {reg: 1}, setup)
    return _fold_setup(setup, reg, ctx)


def _clear(expr):
    for c in expr.children:
        _clear(c)
    if expr.bounds:
        for b in expr.bounds:
            _clear(b)
    expr._res = {}


def _dedup_reads(k):
    """deduplicateReads (passes.nim:352-369)."""
    unique, subs, reads = {}, {}, []
    for (tid, reg, raw, dims) in k.reads:
        key = (tid, raw, tuple((d.constant, tuple(sorted(d.factors.items()))) for d in dims))
        if key in unique:
            subs[reg] = unique[key]
        else:
            unique[key] = reg
            reads.append((tid, reg, raw, dims))
    k.reads = reads
    if subs:
        k.instrs = [(kind, res, [subs.get(a, a) for a in args], extra) for (kind, res, args, extra) in k.instrs]
        k.result = subs.get(k.result, k.result)


def _build_kernel(kb):
    """parser.nim:231-259."""
    _clear(kb.value)
    for d in kb.dims:
        _clear(d)
    ctx = _Ctx()
    k = ctx.kernel
    k.result = _build(kb.value, k.instrs, ctx.block(), ctx)
    wdims = [_build_linear(d, ctx) for d in kb.dims]
    k.write = (_tensor_id(kb.target), k.result, kb.is_raw, wdims)
    _dedup_reads(k)
    if kb.grads is not None:
        k.custom_grad = [_build_kernel(g) for g in kb.grads]
    return k


class Target:
    def __init__(self, name, output):
        self.name, self.output = name, output
        self.kernels = []
        self.shapes = []   # user shape constraints in flatten order (Target.shapes, ir.nim)


class Program:
    def __init__(self):
        self.tensors = []   # dicts: kind, name, shape, range
        self.inputs = {}
        self.targets = {}
        self.scalar = "f32"  # the T of compile[T] (model.nim:253-260): "f32" or "f64"

    def alloc_tensor(self, **kw):
        self.tensors.append(kw)
        return len(self.tensors)

    # ---- kernel-description text (grammar: DESIGN.md) --------------------------------------
    def to_text(self):
        out = ["kd 1 " + self.scalar]
        for i, t in enumerate(self.tensors, 1):
            name = t.get("name") or "-"
            line = ["tensor", str(i), t["kind"], name.replace(" ", "_")]
            shape = t.get("shape")
            if shape is None or (t["kind"] == "input" and not shape):
                line.append("-1")
            else:
                line += [str(len(shape))] + [str(s) for s in shape]
            if t["kind"] in ("param", "random"):
                line += [repr(t["range"][0]), repr(t["range"][1])]
            out.append(" ".join(line))
        # toKd (nim/exprgrad/runtimes/hipmodel.nim): targets in name order; a tensor's shape lines are written once, where
        # the walk over the sorted targets first meets them (a reshape kernel's constraint while its kernel is emitted,
        # a target's user constraints after its kernels); all shape lines precede all targets in the text
        shape_lines, shaped, target_lines = [], set(), []
        for name in sorted(self.targets):
            tgt = self.targets[name]
            # the text is whitespace-separated tokens: a tensor name loses its blanks (above), a target
            # name is the key callers look the target up by and must survive the round trip unchanged
            if not name or any(ch.isspace() for ch in name):
                raise ValueError(f"target name {name!r}: empty or contains whitespace (kernel-description text is token based)")
            target_lines.append(f"target {name} {tgt.output}")
            for k in tgt.kernels:
                if k.generator and k.generator[0] == "reshape":
                    _reshape_text(k.generator, target_lines, shape_lines, shaped)
                elif k.generator:
                    target_lines.append(" ".join(str(x) for x in k.generator))
                else:
                    _kernel_text(k, target_lines)
            target_lines.append("endtarget")
            for c in tgt.shapes:
                if c[1] in shaped:
                    continue
                shaped.add(c[1])
                if c[0] == "copy":
                    shape_lines.append(f"shapecopy {c[1]} {c[2]}")
                else:
                    toks = ["shapedims", str(c[1]), str(len(c[2]))]
                    for d in c[2]:
                        toks += d.tokens()
                    shape_lines.append(" ".join(toks))
                    for ins in (c[3] if len(c) > 3 else ()):
                        shape_lines.append(f"shapesetup {c[1]} " + _ins_text(ins))
        out += shape_lines + target_lines
        return "\n".join(out) + "\n"


def _reshape_text(gen, out, shape_lines, shaped):
    """GenReshape as generate expands it (passes.nim:643-688; emitReshape of hipmodel.nim): a raw copy over len(source)
    — data register 1, iterator 2, len 3 — plus a ShapeDims constraint whose -1 entry is len(source) div the product of
    the others (registers 1, 2, 3 of the constraint)."""
    _, src, dest, dims = gen
    out += ["kernel 3", f"setup len 3 0 {src}", "loop 2 reshape.it 1 L 0 0 L 0 1 3 1", f"read {src} 1 1 1 L 0 1 2 1", "result 1",
            f"write {dest} 1 1 1 L 0 1 2 1", "endkernel"]
    if dest in shaped:
        return
    shaped.add(dest)
    prod = 1
    for size in dims:
        if size >= 0:
            prod *= size
    line, setup = f"shapedims {dest} {len(dims)}", []
    for size in dims:
        if size >= 0:
            line += f" L {size} 0"
        else:
            setup += [f"shapesetup {dest} len 1 0 {src}", f"shapesetup {dest} index 2 0 {prod}", f"shapesetup {dest} indexdiv 3 2 1 2"]
            line += " L 0 1 3 1"
    shape_lines.append(line)
    shape_lines += setup


def _kernel_text(k, out):
    """emitKernel of nim/exprgrad/runtimes/hipmodel.nim, statement for statement: host-evaluated instructions first
    (the setup of explicit loop bounds, then the operand-index instructions that do not depend on an iterator), the
    loops, then the operand-index instructions that do ("idx"), reads, instructions, result, write."""
    out.append(f"kernel {k.nregs}")
    for ins in k.setup:
        out.append("setup " + _ins_text(ins))
    for (reg, nm, bounds) in k.loops:
        if bounds:
            for b in bounds:
                for ins in b.setup:
                    out.append("setup " + _ins_text(ins))
    operand_setup = [ins for (_, _, _, dims) in k.reads for d in dims for ins in d.setup]
    operand_setup += [ins for d in k.write[3] for ins in d.setup]
    varying = {reg for (reg, _, _) in k.loops}
    for (_, res, args, _) in operand_setup:       # dependsOnIterators: in list order, transitively
        if any(a in varying for a in args):
            varying.add(res)
    for ins in operand_setup:
        if ins[1] not in varying:
            out.append("setup " + _ins_text(ins))
    for (reg, nm, bounds) in k.loops:
        if bounds:
            out.append(" ".join(["loop", str(reg), nm, "1"] + bounds[0].tokens() + bounds[1].tokens()))
        else:
            out.append(f"loop {reg} {nm} 0")
    for ins in k.index_instrs:
        out.append("idx " + _ins_text(ins))
    for ins in operand_setup:
        if ins[1] in varying:
            out.append("idx " + _ins_text(ins))
    for (tid, reg, raw, dims) in k.reads:
        toks = ["read", str(tid), str(reg), "1" if raw else "0", str(len(dims))]
        for d in dims:
            toks += d.tokens()
        out.append(" ".join(toks))
    for ins in k.instrs:
        out.append("ins " + _ins_text(ins))
    out.append(f"result {k.result}")
    tid, reg, raw, dims = k.write
    toks = ["write", str(tid), str(reg), "1" if raw else "0", str(len(dims))]
    for d in dims:
        toks += d.tokens()
    out.append(" ".join(toks))
    if k.custom_grad is not None:
        out.append("customgrad")
        for g in k.custom_grad:
            _kernel_text(g, out)
        out.append("endcustomgrad")
    out.append("endkernel")


def _ins_text(ins):
    kind, res, args, extra = ins
    toks = [kind, str(res), str(len(args))] + [str(a) for a in args]
    if kind == "scalar":
        toks.append(repr(float(extra)))
    elif kind == "index":
        toks.append(str(int(extra)))
    elif kind == "boolean":
        toks.append("1" if extra else "0")
    elif kind == "shape":
        toks += [str(extra[0]), str(extra[1])]
    elif kind in ("len", "shapelen"):
        toks.append(str(extra[0]))
    return " ".join(toks)


def _alloc_tensors(fun, program):
    """parser.nim:261-315."""
    if fun.tensor == 0:
        k = fun.kind
        if k == "input":
            if fun.name not in program.inputs:
                program.inputs[fun.name] = program.alloc_tensor(kind="input", name=fun.name, shape=list(fun.input_shape))
            fun.tensor = program.inputs[fun.name]
            if program.tensors[fun.tensor - 1]["shape"] != list(fun.input_shape):
                raise ParserError(f'Expected shapes for input "{fun.name}" do not match.')
        elif k == "param":
            fun.tensor = program.alloc_tensor(kind="param", name=fun.name, shape=list(fun.param_shape), range=fun.init_range)
        elif k in ("result", "gradient", "reshape"):
            fun.tensor = program.alloc_tensor(kind="result", name=fun.name)
        elif k == "random":                                  # parser.nim:281-285
            fun.tensor = program.alloc_tensor(kind="random", name=fun.name, range=fun.random_range)
        elif k == "effect":
            _alloc_tensors(fun.effect, program)
            fun.tensor = fun.effect.tensor
        elif k == "cache":
            # TensorCache (parser.nim:296-303): shaped like the tensor it shadows (a parameter)
            _alloc_tensors(fun.cache, program)
            src = program.tensors[fun.cache.tensor - 1]
            if src.get("shape") is None:
                raise ParserError("cache() needs a tensor with a static shape (a parameter)")
            fun.tensor = program.alloc_tensor(kind="cache", name=fun.name, shape=list(src["shape"]))
        elif k == "cond":                                    # parser.nim:302-307
            for child in fun.cond.values():
                _alloc_tensors(child, program)
            if fun.cond_else is not None:
                _alloc_tensors(fun.cond_else, program)
        for c in fun.children:
            _alloc_tensors(c, program)
        if k == "target":
            fun.tensor = fun.children[0].tensor


def _flatten(fun, target, program):
    """parser.nim:317-384."""
    if target.name in fun.targets:
        return
    for c in fun.children:
        _flatten(c, target, program)
    if fun.kind == "effect":
        _flatten(fun.effect, target, program)
    fun.targets.add(target.name)
    if fun.kind in ("result", "effect"):
        for kb in fun.kernels:
            target.kernels.append(_build_kernel(kb))
        if fun.shape_constr:
            if fun.shape_constr[0] == "copy":
                c = ("copy", fun.tensor, fun.shape_constr[1].tensor)
            else:
                # ShapeDims (parser.nim:345-357): every dimension is built with a register file of its own and never
                # folded; the text shares one "shapesetup" list per constraint, so the registers of dimension d are
                # shifted by the highest register of the dimensions before it (emitShapeConstraint, hipmodel.nim)
                dims, setup, offset = [], [], 0
                for d in fun.shape_constr[1]:
                    _clear(d)
                    lin = _build_linear(d, _Ctx(), fold=False)
                    highest = 0
                    for (kind, res, args, extra) in lin.setup:
                        setup.append((kind, res + offset, [a + offset for a in args], extra))
                        highest = res if res > highest else highest
                    for r in lin.factors:
                        highest = r if r > highest else highest
                    dims.append(LinearIndex(lin.constant, {r + offset: f for r, f in lin.factors.items()}))
                    offset += highest
                c = ("dims", fun.tensor, dims, tuple(setup))
            target.shapes.append(c)
    elif fun.kind == "random":                               # parser.nim:378-383: shaped like its argument
        target.shapes.append(("copy", fun.tensor, fun.children[0].tensor))
    elif fun.kind == "reshape":                              # parser.nim:358-367: GenReshape
        k = Kernel()
        k.generator = ("reshape", fun.children[0].tensor, fun.tensor, tuple(fun.reshape_dims))
        target.kernels.append(k)
    elif fun.kind == "cond":                                 # parser.nim:368-377
        child = fun.cond.get(target.name, fun.cond_else)
        if child is None:
            raise ParserError(f'Conditional node does not have a branch for the target "{target.name}"')
        _flatten(child, target, program)
        fun.tensor = child.tensor
    elif fun.kind == "backwards":
        k = Kernel()
        k.generator = ("backwards", fun.children[0].tensor)
        target.kernels.append(k)
    elif fun.kind == "gradient":
        k = Kernel()
        k.generator = ("gradient", fun.children[1].tensor, fun.tensor)
        target.kernels.append(k)


def _collect_targets(fun, targets):
    """parser.nim:386-402."""
    if fun.kind == "target":
        if fun.name in targets:
            if targets[fun.name] is not fun:
                raise ParserError(f'There are multiple targets named "{fun.name}". Target names must be unique '
                                  "within a model. Choose a different name every target.")
            return
        targets[fun.name] = fun
    if fun.kind == "cond":                                   # parser.nim:395-399
        for child in fun.cond.values():
            _collect_targets(child, targets)
        if fun.cond_else is not None:
            _collect_targets(fun.cond_else, targets)
    for c in fun.children:
        _collect_targets(c, targets)
    if fun.kind == "effect":
        _collect_targets(fun.effect, targets)


def to_program(*graphs):
    """toProgram (parser.nim:404-417)."""
    program = Program()
    targets = {}
    for fun in graphs:
        _alloc_tensors(fun, program)
        _collect_targets(fun, targets)
    for name, fun in targets.items():
        t = Target(name, fun.tensor)
        _flatten(fun, t, program)
        program.targets[name] = t
    return program


toProgram = to_program
