"""ORACLE (test infrastructure) — graph-level restatement of exprgrad's pipeline for the hot path.

Takes the same kernel-description text the product consumes (the reference's `Program` before
`generate`), and does what `model.compile` + `Model.call` do on the CPU target:

    generate (autodiff)            passes.nim:383-549 (derive), 558-640 (GenBackwards / GenGradient)
    deadKernelElim                 passes.nim:331-350
    inferLoopBounds                passes.nim:986-1010 (first tensor op indexed by the bare iterator)
    run-time shape inference       passes.nim:1059-1095 (constraints), 1386-1436 (solve) — restated as a
                                   forward walk over the kernel list, which is what the sorted
                                   constraints amount to for these programs
    reorderLoops                   passes.nim:700-745 (reads weigh 10, writes 1)
    call: zero results, run kernels in order, return the output      model.nim:275-300, 385-406

The arithmetic is executed by oracle/refinterp.c (one lowered kernel at a time, sequential f32
accumulation in loop order).  Written independently of the product's C++ implementation
(exprgrad_amd/csrc/host): the two share only the input text.
"""
import ctypes
import math

import numpy as np

from . import refcpu

SCALAR, INDEX, BOOLEAN = "scalar", "index", "boolean"


class Lin:
    def __init__(self, constant=0, factors=None):
        self.constant = constant
        self.factors = dict(factors or {})

    def only_register(self):
        """onlyRegister (passes.nim:968-972)."""
        if self.constant == 0 and len(self.factors) == 1:
            (r, f), = self.factors.items()
            if f == 1:
                return r
        return 0

    def key(self):
        return (self.constant, tuple(sorted(self.factors.items())))


class Op:
    def __init__(self, tensor, reg, raw, dims):
        self.tensor, self.reg, self.raw, self.dims = tensor, reg, raw, dims

    def key(self):
        return (self.tensor, self.raw, tuple(d.key() for d in self.dims))


class Instr:
    def __init__(self, kind, res, args, extra=None):
        self.kind, self.res, self.args, self.extra = kind, res, list(args), extra


class Loop:
    def __init__(self, reg, name, bounds=None):
        self.reg, self.name, self.bounds = reg, name, bounds


class Kernel:
    def __init__(self):
        self.nregs = 0
        self.setup, self.loops, self.reads, self.instrs = [], [], [], []
        self.index_instrs = []   # Index instructions evaluated inside the loop nest before the reads
                                 # (LinearIndex.setup of the reference, ir.nim:120-123): `y div 2`
        self.result = 0
        self.write = None
        self.generator = None
        self.is_seed = False
        self.custom_grad = None  # KernelGradient(isCustom) ir.nim:203-209: kernels; tensor -t = grad of t

    def alloc(self):
        self.nregs += 1
        return self.nregs

    def clone(self):
        k = Kernel()
        k.nregs = self.nregs
        k.setup = list(self.setup)
        k.loops = list(self.loops)
        k.reads = list(self.reads)
        k.instrs = list(self.instrs)
        k.index_instrs = list(self.index_instrs)
        k.result, k.write, k.generator = self.result, self.write, self.generator
        k.is_seed = self.is_seed
        k.custom_grad = self.custom_grad
        return k


class Program:
    def __init__(self):
        self.tensors = {}       # id -> dict(kind, name, shape, range)
        self.shape_copy = {}    # dest -> src
        self.shape_dims = {}    # dest -> [Lin]
        self.shape_setup = {}   # dest -> [Instr] evaluated on the host for the Lin registers of shape_dims
        self.targets = {}       # name -> (output, [Kernel])
        self.inputs = {}


# ---------------------------------------------------------------------------------- parsing

def _lin(toks, pos):
    assert toks[pos] == "L", toks[pos:]
    c, n = int(toks[pos + 1]), int(toks[pos + 2])
    pos += 3
    factors = {}
    for _ in range(n):
        factors[int(toks[pos])] = int(toks[pos + 1])
        pos += 2
    return Lin(c, factors), pos


def _op(toks):
    tid, reg, raw, nd = int(toks[1]), int(toks[2]), toks[3] == "1", int(toks[4])
    pos, dims = 5, []
    for _ in range(nd):
        d, pos = _lin(toks, pos)
        dims.append(d)
    return Op(tid, reg, raw, dims)


def _instr(toks):
    kind, res, n = toks[0], int(toks[1]), int(toks[2])
    args = [int(t) for t in toks[3:3 + n]]
    rest = toks[3 + n:]
    extra = None
    if kind == "scalar":
        extra = float(rest[0])
    elif kind == "index":
        extra = int(rest[0])
    elif kind == "boolean":
        extra = rest[0] == "1"
    elif kind == "shape":
        extra = (int(rest[0]), int(rest[1]))
    elif kind in ("len", "shapelen"):
        extra = (int(rest[0]),)
    return Instr(kind, res, args, extra)


def parse(text):
    prog = Program()
    cur_target = None
    cur = None
    stack = []   # kernels whose customgrad block is open
    for line in text.splitlines():
        toks = line.split()
        if not toks or toks[0].startswith("#"):
            continue
        t = toks[0]
        if t == "kd":
            assert toks[1] == "1" and toks[2] in ("f32", "f64")
            prog.scalar = toks[2]
        elif t == "tensor":
            tid, kind, name, rank = int(toks[1]), toks[2], toks[3], int(toks[4])
            d = {"kind": kind, "name": "" if name == "-" else name, "shape": None}
            if rank >= 0:
                d["shape"] = [int(x) for x in toks[5:5 + rank]]
            if kind in ("param", "random"):
                d["range"] = (float(toks[5 + max(rank, 0)]), float(toks[6 + max(rank, 0)]))
            prog.tensors[tid] = d
            if kind == "input":
                prog.inputs[d["name"]] = tid
        elif t == "shapecopy":
            prog.shape_copy[int(toks[1])] = int(toks[2])
        elif t == "shapedims":
            n, pos, dims = int(toks[2]), 3, []
            for _ in range(n):
                d, pos = _lin(toks, pos)
                dims.append(d)
            prog.shape_dims[int(toks[1])] = dims
        elif t == "shapesetup":
            prog.shape_setup.setdefault(int(toks[1]), []).append(_instr(toks[2:]))
        elif t == "target":
            cur_target = (int(toks[2]), [])
            prog.targets[toks[1]] = cur_target
        elif t == "endtarget":
            cur_target = None
        elif t == "kernel":
            cur = Kernel()                 # inside an open customgrad block: a gradient kernel of stack[-1]
            cur.nregs = int(toks[1])
        elif t == "endkernel":
            if stack:
                stack[-1].custom_grad.append(cur)
                cur = None
            else:
                cur_target[1].append(cur)
                cur = None
        elif t == "customgrad":
            cur.custom_grad = []
            stack.append(cur)
            cur = None
        elif t == "endcustomgrad":
            cur = stack.pop()
        elif t == "idx":
            cur.index_instrs.append(_instr(toks[1:]))
        elif t == "setup":
            cur.setup.append(_instr(toks[1:]))
        elif t == "loop":
            reg, name, has = int(toks[1]), toks[2], toks[3] == "1"
            bounds = None
            if has:
                a, pos = _lin(toks, 4)
                b, pos = _lin(toks, pos)
                bounds = (a, b)
            cur.loops.append(Loop(reg, name, bounds))
        elif t == "read":
            cur.reads.append(_op(toks))
        elif t == "ins":
            cur.instrs.append(_instr(toks[1:]))
        elif t == "result":
            cur.result = int(toks[1])
        elif t == "write":
            cur.write = _op(toks)
            # (the last statement of a kernel body) index instructions that do not change inside the loop nest are host
            # values — the Nim emitter writes them as "setup" already (hipmodel.nim emitKernel); text from another
            # producer is normalised the same way
            varying, stay = {lp.reg for lp in cur.loops}, []
            for ins in cur.index_instrs:
                if any(a in varying for a in ins.args):
                    varying.add(ins.res)
                    stay.append(ins)
                else:
                    cur.setup.append(ins)
            cur.index_instrs = stay
        elif t == "backwards":
            k = Kernel()
            k.generator = ("backwards", int(toks[1]))
            cur_target[1].append(k)
        elif t == "gradient":
            k = Kernel()
            k.generator = ("gradient", int(toks[1]), int(toks[2]))
            cur_target[1].append(k)
        else:
            raise ValueError("unknown statement: " + line)
    return prog


# ---------------------------------------------------------------------------------- autodiff

def derive_instrs(instrs, k, grad_regs):
    """derive(instrs, regs, gradRegs)  passes.nim:383-517.  Instructions are visited last to first."""
    out = []

    def emit(kind, args, extra=None):
        r = k.alloc()
        out.append(Instr(kind, r, args, extra))
        return r

    for ins in reversed(instrs):
        if ins.res not in grad_regs:
            continue
        g = grad_regs[ins.res]
        a = ins.args
        kind = ins.kind
        if kind == "add":                                   # 393-394
            ga = [g, g]
        elif kind == "sub":                                 # 395-398
            ga = [g, emit("negate", [g])]
        elif kind == "mul":                                 # 399-403
            g_a = emit("mul", [g, a[1]])
            g_b = emit("mul", [g, a[0]])
            ga = [g_a, g_b]
        elif kind == "div":                                 # 404-415 (emission order as in the source)
            g_a = emit("div", [g, a[1]])
            sq_y = emit("mul", [a[1], a[1]])
            div_g = emit("div", [g, sq_y])
            neg_x = emit("negate", [a[0]])
            g_b = emit("mul", [neg_x, div_g])
            ga = [g_a, g_b]
        elif kind == "negate":                              # 416-419
            ga = [emit("negate", [g])]
        elif kind in ("ln", "log10", "log2"):               # 420-436
            base = {"ln": 1.0, "log10": math.log(10.0), "log2": math.log(2.0)}[kind]
            den = a[0]
            if base != 1.0:
                factor = emit("scalar", [], base)
                den = emit("mul", [a[0], factor])
            ga = [emit("div", [g, den])]
        elif kind == "log":                                 # 437-455
            log_y = emit("ln", [a[1]])
            mul = emit("mul", [a[0], log_y])
            g_x = emit("div", [g, mul])
            log_x = emit("ln", [a[0]])
            neg_log_x = emit("negate", [log_x])
            log_y_sq = emit("mul", [log_y, log_y])
            den = emit("mul", [a[1], log_y_sq])
            num = emit("mul", [g, neg_log_x])
            g_y = emit("div", [num, den])
            ga = [g_x, g_y]
        elif kind == "exp":                                 # 456-459: reuses the forward result register
            ga = [emit("mul", [g, ins.res])]
        elif kind == "sin":                                 # 460-464
            c = emit("cos", [a[0]])
            ga = [emit("mul", [c, g])]
        elif kind == "cos":                                 # 465-470
            s = emit("sin", [a[0]])
            ns = emit("negate", [s])
            ga = [emit("mul", [ns, g])]
        elif kind == "select":                              # 471-476
            zero = emit("scalar", [], 0.0)
            g_a = emit("select", [a[0], g, zero])
            g_b = emit("select", [a[0], zero, g])
            ga = [0, g_a, g_b]
        elif kind == "sqrt":                                # 477-484
            two = emit("scalar", [], 2.0)
            den = emit("mul", [two, ins.res])
            ga = [emit("div", [g, den])]
        elif kind == "pow":                                 # 485-503
            one = emit("scalar", [], 1.0)
            new_exp = emit("sub", [a[1], one])
            pw = emit("pow", [a[0], new_exp])
            pw_factor = emit("mul", [a[1], pw])
            g_base = emit("mul", [g, pw_factor])
            lg = emit("ln", [a[0]])
            product = emit("mul", [ins.res, lg])
            g_exp = emit("mul", [g, product])
            ga = [g_base, g_exp]
        elif kind in ("toscalar", "toindex"):               # 504
            ga = [0]
        else:
            ga = []
        if len(ga) != len(a):
            raise ValueError("Unable to derive " + kind)    # 507-508
        for arg, garg in zip(a, ga):                        # 510-517
            if garg != 0:
                if arg in grad_regs:
                    grad_regs[arg] = emit("add", [grad_regs[arg], garg])
                else:
                    grad_regs[arg] = garg
    return out


def dead_code_elim(k):
    """Kernel-level deadCodeElim (passes.nim:306-317): keep what the write needs."""
    used = set()
    if k.write.reg:
        used.add(k.write.reg)
    for d in k.write.dims:
        used.update(d.factors)
    kept = []
    for ins in reversed(k.instrs):
        if ins.res in used:
            kept.append(ins)
            used.update(ins.args)
    k.instrs = kept[::-1]
    reads = []
    for r in k.reads:
        if r.reg in used:
            reads.append(r)
            for d in r.dims:
                used.update(d.factors)
    k.reads = reads
    kept_idx = []
    for ins in reversed(k.index_instrs):
        if ins.res in used:
            kept_idx.append(ins)
            used.update(ins.args)
    k.index_instrs = kept_idx[::-1]
    # loops are kept only if their iterator is used (passes.nim:280-289)
    loops = []
    for lp in k.loops:
        if lp.reg in used:
            loops.append(lp)
            if lp.bounds:
                for b in lp.bounds:
                    used.update(b.factors)
    k.loops = loops
    k.setup = [s for s in k.setup if s.res in used]


def derive_kernel(kernel, grad_tensors):
    """derive(kernel, gradTensors)  passes.nim:519-549: one kernel per read, in read order."""
    base = kernel.clone()
    grad_regs = {}
    write_grad = base.alloc()
    base.reads.append(Op(grad_tensors[kernel.write.tensor], write_grad, kernel.write.raw, kernel.write.dims))
    grad_regs[kernel.write.reg] = write_grad
    base.instrs = base.instrs + derive_instrs(kernel.instrs, base, grad_regs)
    out = []
    for read in kernel.reads:
        if read.reg in grad_regs:
            gk = base.clone()
            gk.result = grad_regs[read.reg]
            gk.write = Op(grad_tensors[read.tensor], grad_regs[read.reg], read.raw, read.dims)
            dead_code_elim(gk)
            out.append(gk)
    return out


def generate(prog, kernels):
    """generate (passes.nim:558-640) for GenBackwards / GenGradient."""
    kernels = list(kernels)
    i = 0
    while i < len(kernels):
        k = kernels[i]
        if k.generator and k.generator[0] == "backwards":
            grad_tensors = {}
            grad_kernels = []
            loss = k.generator[1]
            grad_loss = _alloc_tensor(prog, "result", "grad_loss")
            prog.shape_copy[grad_loss] = loss
            seed = Kernel()            # gradLoss{i} = 1 for i in 0 ..< len(loss)   (575-606)
            r_val, r_it, r_len = seed.alloc(), seed.alloc(), seed.alloc()
            seed.setup = [Instr("len", r_len, [], (loss,))]
            seed.loops = [Loop(r_it, "i", (Lin(0), Lin(0, {r_len: 1})))]
            seed.instrs = [Instr("scalar", r_val, [], 1.0)]
            seed.result = r_val
            seed.write = Op(grad_loss, r_val, True, [Lin(0, {r_it: 1})])
            seed.is_seed = True
            grad_kernels.append(seed)
            grad_tensors[loss] = grad_loss
            for k2 in kernels[i + 1:]:                      # 608-612
                if k2.generator and k2.generator[0] == "gradient":
                    grad_tensors[k2.generator[1]] = k2.generator[2]
                    prog.shape_copy[k2.generator[2]] = k2.generator[1]
            for k2 in reversed(kernels[:i]):                # 614-636
                if k2.generator:
                    continue
                for read in k2.reads:
                    if read.tensor not in grad_tensors:
                        gt = _alloc_tensor(prog, "result", "grad")
                        prog.shape_copy[gt] = read.tensor
                        grad_tensors[read.tensor] = gt
                if k2.write.tensor not in grad_tensors:
                    continue  # does not influence the loss (the reference would raise KeyError)
                if k2.custom_grad is not None:              # 626-634: the user's kernels, last first
                    for gk in reversed(k2.custom_grad):
                        grad_kernels.append(_substitute_grads(gk, grad_tensors))
                else:
                    grad_kernels.extend(derive_kernel(k2, grad_tensors))
            kernels[i:i + 1] = grad_kernels
            i += len(grad_kernels)
        elif k.generator and k.generator[0] == "gradient":
            del kernels[i]                                  # 641-642
        else:
            i += 1
    return kernels


def _substitute_grads(kernel, grad_tensors):
    """A customGrad kernel with its gradient placeholders (tensor -t) bound to gradTensors[t]."""
    def sub(op):
        return Op(grad_tensors[-op.tensor], op.reg, op.raw, op.dims) if op.tensor < 0 else op
    gk = kernel.clone()
    gk.custom_grad = None
    gk.reads = [sub(r) for r in gk.reads]
    gk.write = sub(gk.write)
    return gk


def _alloc_tensor(prog, kind, name):
    tid = max(prog.tensors) + 1 if prog.tensors else 1
    prog.tensors[tid] = {"kind": kind, "name": name, "shape": None}
    return tid


def dead_kernel_elim(prog, output, kernels):
    """passes.nim:331-350."""
    used = {tid for tid, t in prog.tensors.items() if t["kind"] != "result"}
    if output:
        used.add(output)
    kept = []
    for k in reversed(kernels):
        if k.write.tensor in used:
            for r in k.reads:
                used.add(r.tensor)
            kept.append(k)
    return kept[::-1]


def reorder_loops(k):
    """reorderLoops (passes.nim:700-745)."""
    loop_of = {lp.reg: i for i, lp in enumerate(k.loops)}
    n = len(k.loops)
    graph = [{"read": [], "write": []} for _ in range(n)]
    ops = [("read", r) for r in k.reads] + [("write", k.write)]
    for kind, op in ops:
        for d in range(1, len(op.dims)):
            for ra in op.dims[d - 1].factors:
                for rb in op.dims[d].factors:
                    if ra in loop_of and rb in loop_of:
                        graph[loop_of[ra]][kind].append(loop_of[rb])
    score_vals = {"read": 10, "write": 1}
    scores = [0] * n
    for edges in graph:
        for kind, tg in edges.items():
            for t in tg:
                scores[t] += score_vals[kind]
    closed = [False] * n
    order = []
    for _ in range(n):
        best, best_score = -1, 0
        for i, s in enumerate(scores):
            if not closed[i] and (s < best_score or best < 0):
                best, best_score = i, s
        closed[best] = True
        order.append(best)
        for kind, tg in graph[best].items():
            for t in tg:
                scores[t] -= score_vals[kind]
    k.loops = [k.loops[i] for i in order]


def compile_target(prog, name):
    output, kernels = prog.targets[name]
    grads = [(k.generator[1], k.generator[2]) for k in kernels
             if k.generator and k.generator[0] == "gradient" and prog.tensors[k.generator[1]]["kind"] == "param"]
    prog.param_grads = getattr(prog, "param_grads", {})
    prog.param_grads[name] = grads
    all_kernels = generate(prog, kernels)
    # Shape constraints are collected from every kernel BEFORE dead kernels are dropped
    # (model.nim:46-77: inferShapeConstraints precedes generate/deadKernelElim), so the
    # eliminated kernels still take part in shape inference.
    live = dead_kernel_elim(prog, output, all_kernels)
    for k in live:
        reorder_loops(k)
    return output, live, all_kernels


# ---------------------------------------------------------------------------------- run time

class ShapeError(Exception):
    pass


def _sdiv(a, b):
    if b == 0:
        return 0
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _eval_setup(k, shapes, epoch):
    return _eval_host_instrs(k.setup, shapes, epoch)


def _user_shape(prog, tid, shapes, epoch, kernel_vals):
    """withShape / reshape dims of `tid` if they can be evaluated now, else None.  Registers of the
    dims come from the constraint's own host instructions (shapesetup) or, failing that, from the
    writing kernel's setup."""
    dims = prog.shape_dims.get(tid)
    if dims is None:
        return None
    vals = dict(kernel_vals)
    try:
        vals.update(_eval_host_instrs(prog.shape_setup.get(tid, []), shapes, epoch))
        return [_lin_const(d, vals) for d in dims]
    except (KeyError, TypeError):
        return None


def _eval_host_instrs(instrs, shapes, epoch, later_tensor=None, later=None):
    """later_tensor / later: instructions that need the shape of `later_tensor` while it is unknown,
    and everything computed from them, are skipped; their registers are collected in `later`."""
    vals = {}
    for s in instrs:
        if later is not None:
            needs = (s.kind in ("shape", "len", "shapelen") and
                     (s.extra[0] if isinstance(s.extra, (tuple, list)) else s.extra) == later_tensor and
                     shapes.get(later_tensor) is None)
            if needs or any(r in later for r in s.args):
                later.add(s.res)
                continue
        if s.kind in ("add", "sub", "mul", "indexdiv", "mod", "negate"):
            a = [vals[r] for r in s.args]
            if s.kind == "add":
                vals[s.res] = a[0] + a[1]
            elif s.kind == "sub":
                vals[s.res] = a[0] - a[1]
            elif s.kind == "mul":
                vals[s.res] = a[0] * a[1]
            elif s.kind == "negate":
                vals[s.res] = -a[0]
            elif s.kind == "indexdiv":           # sdiv (llvmgen.nim:236): truncation toward zero
                vals[s.res] = _sdiv(a[0], a[1])
            else:                                # srem
                vals[s.res] = a[0] - _sdiv(a[0], a[1]) * a[1]
        elif s.kind == "shape":
            tid, dim = s.extra
            vals[s.res] = shapes[tid][dim]
        elif s.kind == "len":
            vals[s.res] = int(np.prod(shapes[s.extra[0]], dtype=np.int64))
        elif s.kind == "shapelen":
            vals[s.res] = len(shapes[s.extra[0]])
        elif s.kind == "index":
            vals[s.res] = s.extra
        elif s.kind == "epoch":
            vals[s.res] = epoch
        else:
            raise ValueError("setup instruction " + s.kind)
    return vals


def _lin_const(lin, vals):
    v = lin.constant
    for r, f in lin.factors.items():
        v += f * vals[r]
    return v


def infer_kernel(prog, k, shapes, epoch=0):
    """Loop bounds (inferLoopBounds, passes.nim:986-1010) and the write tensor's shape
    (inferShapeConstraints, passes.nim:1059-1095 + the linear solve of 1420-1436)."""
    # explicit loop bounds may name the shape of the very tensor the kernel writes
    # (`res[x] ++= ... | (x in 0..<res.shape[0])`, tests/test_model.nim:99-107): the reference's
    # constraint solver gets that shape from the reads; such bounds are applied once it is known
    later = set()
    vals = _eval_host_instrs(k.setup, shapes, epoch, k.write.tensor, later)
    bounds = {}
    postponed = set()
    for lp in k.loops:
        if not lp.bounds:
            continue
        if any(r in later for b in lp.bounds for r in b.factors):
            postponed.add(lp.reg)
        else:
            bounds[lp.reg] = (_lin_const(lp.bounds[0], vals), _lin_const(lp.bounds[1], vals))
    # user constraints (withShape / copyShape, parser.nim:683-697) fix the written tensor's shape
    # before its loops are bounded (PriorityUser outranks inferred constraints)
    wt0 = k.write.tensor
    if shapes.get(wt0) is None:
        user = _user_shape(prog, wt0, shapes, epoch, vals)
        if user is not None:
            shapes[wt0] = user
        elif wt0 in prog.shape_copy and shapes.get(prog.shape_copy[wt0]) is not None:
            shapes[wt0] = list(shapes[prog.shape_copy[wt0]])
    ops = list(k.reads) + [k.write]
    idx_regs = {i.res for i in k.index_instrs}   # computed indices never bound a loop
    for op in ops:
        shp = shapes.get(op.tensor)
        if shp is None:
            continue
        if not op.raw and len(op.dims) != len(shp):
            raise ShapeError(f"tensor {op.tensor} has rank {len(shp)} but is indexed with {len(op.dims)} dims")
        for d, lin in enumerate(op.dims):
            r = lin.only_register()
            # (loops with explicit bounds never take part: inferLoopBounds skips them, passes.nim:1030-1038)
            if r and r not in bounds and r not in postponed and any(lp.reg == r for lp in k.loops):
                bounds[r] = (0, int(np.prod(shp, dtype=np.int64)) if op.raw else shp[d])
    # iterators that never appear bare: solve  sum f*(max iter) + c = dim - 1  (valid convolution)
    progress = True
    while progress and any(lp.reg not in bounds for lp in k.loops):
        progress = False
        for op in ops:
            shp = shapes.get(op.tensor)
            if shp is None or op.raw:
                continue
            for d, lin in enumerate(op.dims):
                if any(r in idx_regs for r in lin.factors):
                    continue
                unknown = [r for r in lin.factors if r not in bounds and r not in vals]
                if len(unknown) == 1 and lin.factors[unknown[0]] > 0:
                    rest = lin.constant
                    if op is not k.write:
                        # reads of one tensor that differ only in the constant (image[x], image[x + 1],
                        # image[x + 2]): the largest offset decides (simplifyMaxIndex, passes.nim:1040-1057)
                        for other in k.reads:
                            if (other.tensor == op.tensor and not other.raw and len(other.dims) == len(op.dims)
                                    and other.dims[d].factors == lin.factors):
                                rest = max(rest, other.dims[d].constant)
                    for r, f in lin.factors.items():
                        if r == unknown[0]:
                            continue
                        if r in vals:
                            rest += f * vals[r]
                        else:
                            rest += f * ((bounds[r][1] - 1) if f > 0 else bounds[r][0])
                    f = lin.factors[unknown[0]]
                    bounds[unknown[0]] = (0, (shp[d] - 1 - rest) // f + 1)
                    progress = True
    for lp in k.loops:
        if lp.reg not in bounds:
            raise ShapeError(f"unable to infer bounds of loop '{lp.name}'")
    # shape of the written tensor
    wt = k.write.tensor
    if shapes.get(wt) is None:
        if wt in prog.shape_dims:
            user = _user_shape(prog, wt, shapes, epoch, vals)
            if user is None:
                raise ShapeError(f"withShape of tensor {wt} uses a value that is not known yet")
            shapes[wt] = user
        elif wt in prog.shape_copy and shapes.get(prog.shape_copy[wt]) is not None:
            shapes[wt] = list(shapes[prog.shape_copy[wt]])
        elif k.write.raw:
            if len(k.reads) == 1:                          # ShapeCopy, passes.nim:1061-1067
                shapes[wt] = list(shapes[k.reads[0].tensor])
            else:
                raise ShapeError(f"shape of tensor {wt} is under-constrained (raw write, {len(k.reads)} reads)")
        else:
            shp = []
            for lin in k.write.dims:
                if any(r in idx_regs for r in lin.factors):
                    raise ShapeError(f"shape of tensor {wt} is under-constrained (computed write index; use withShape)")
                hi = lin.constant
                for r, f in lin.factors.items():
                    if r in vals:
                        hi += f * vals[r]
                    else:
                        hi += f * ((bounds[r][1] - 1) if f > 0 else bounds[r][0])
                shp.append(hi + 1)
            shapes[wt] = shp
    if later:
        vals = _eval_setup(k, shapes, epoch)
        for lp in k.loops:
            if lp.bounds:
                bounds[lp.reg] = (_lin_const(lp.bounds[0], vals), _lin_const(lp.bounds[1], vals))
    return bounds, vals


_OPC = {
    "scalar": 0, "index": 1, "boolean": 2,
    ("add", SCALAR): 10, ("sub", SCALAR): 11, ("mul", SCALAR): 12, ("div", SCALAR): 13, ("negate", SCALAR): 14,
    ("add", INDEX): 20, ("sub", INDEX): 21, ("mul", INDEX): 22, "indexdiv": 23, "mod": 24, "wrap": 25,
    ("negate", INDEX): 26,
    "sin": 30, "cos": 31, "exp": 32, "pow": 33, "sqrt": 34, "log": 35, "log10": 36, "log2": 37, "ln": 38,
    ("eq", SCALAR): 40, ("lt", SCALAR): 41, ("le", SCALAR): 42, ("eq", INDEX): 43, ("lt", INDEX): 44,
    ("le", INDEX): 45, ("eq", BOOLEAN): 43, "and": 46, "or": 47, "select": 50, "toscalar": 60, "toindex": 61,
}


def infer_types(k, vals):
    """inferTypes restated: register types from their producers."""
    typ = {}
    for lp in k.loops:
        typ[lp.reg] = INDEX
    for r in vals:
        typ[r] = INDEX
    for rd in k.reads:
        typ[rd.reg] = SCALAR
    for ins in list(k.index_instrs) + list(k.instrs):
        kd = ins.kind
        if kd == "scalar":
            t = SCALAR
        elif kd in ("index", "shape", "len", "shapelen", "epoch", "indexdiv", "mod", "wrap", "toindex"):
            t = INDEX
        elif kd in ("boolean", "eq", "lt", "le", "and", "or"):
            t = BOOLEAN
        elif kd in ("add", "sub", "mul", "negate"):
            t = typ[ins.args[0]]
        elif kd == "select":
            t = typ[ins.args[1]]
        else:
            t = SCALAR
        typ[ins.res] = t
    return typ


def _encode_instrs(instrs, typ, shapes, epoch):
    """Instruction list -> (5 x int32 words, literal per instruction) for refinterp.c."""
    words, lits = [], []
    for ins in instrs:
        kd = ins.kind
        lit = 0.0
        if kd in ("shape", "len", "shapelen", "epoch"):
            # host-evaluated builtins (model.nim:83-104) become index literals
            if kd == "shape":
                lit = shapes[ins.extra[0]][ins.extra[1]]
            elif kd == "len":
                lit = int(np.prod(shapes[ins.extra[0]], dtype=np.int64))
            elif kd == "shapelen":
                lit = len(shapes[ins.extra[0]])
            else:
                lit = epoch
            code = 1
        elif kd in ("scalar", "index", "boolean"):
            code = _OPC[kd]
            lit = float(ins.extra)
        elif (kd, typ[ins.args[0]] if ins.args else None) in _OPC:
            code = _OPC[(kd, typ[ins.args[0]])]
        else:
            code = _OPC[kd]
        a = ins.args + [0, 0, 0]
        words += [code, ins.res, a[0], a[1], a[2]]
        lits.append(lit)
    return words, lits


def _run_kernel_indexed(k, bounds, vals, shapes, tensors, epoch, f64=False):
    """Kernels with computed (non-affine) indices: flat index of an operand = constant +
    sum(coefficient * register) over iterator AND index-instruction registers (ref_interp_kernel2)."""
    lib = refcpu.lib()
    interp = {False: lib.ref_interp_kernel2, True: lib.ref_interp_kernel2_f64, "c64": lib.ref_interp_kernel2_c64}[f64]
    interp.restype = ctypes.c_int
    typ = infer_types(k, vals)
    c_i64, c_i32 = ctypes.c_int64, ctypes.c_int32
    nl = len(k.loops)

    def terms(op):
        shp = shapes[op.tensor]
        if op.raw:
            strides = [1]
        else:
            strides = [1] * len(shp)
            for d in range(len(shp) - 2, -1, -1):
                strides[d] = strides[d + 1] * shp[d + 1]
        const, coef = 0, {}
        for d, lin in enumerate(op.dims):
            const += strides[d] * lin.constant
            for r, f in lin.factors.items():
                if r in vals:
                    const += strides[d] * f * vals[r]
                else:
                    coef[r] = coef.get(r, 0) + strides[d] * f
        out = [const, len(coef)]
        for r, f in coef.items():
            out += [r, f]
        return out

    packed, offsets = [], []
    for op in list(k.reads) + [k.write]:
        offsets.append(len(packed))
        packed += terms(op)
    # host values (setup registers) an index instruction may name enter the interpreter's register file as literals
    host = [Instr("index", r, [], int(v)) for r, v in sorted(vals.items())]
    idx_all = host + list(k.index_instrs)
    idx_words, idx_lits = _encode_instrs(idx_all, typ, shapes, epoch)
    words, lits = _encode_instrs(k.instrs, typ, shapes, epoch)
    nreads = len(k.reads)
    out = tensors[k.write.tensor]
    arr = lambda t, v: (t * max(len(v), 1))(*v)
    rc = interp(
        nl, arr(c_i64, [bounds[lp.reg][0] for lp in k.loops]), arr(c_i64, [bounds[lp.reg][1] for lp in k.loops]),
        arr(c_i32, [lp.reg for lp in k.loops]), k.nregs + 1,
        len(idx_all), arr(c_i32, idx_words), arr(ctypes.c_double, idx_lits),
        nreads, arr(ctypes.c_void_p, [tensors[r.tensor].ctypes.data for r in k.reads]),
        arr(c_i32, [r.reg for r in k.reads]), arr(c_i64, packed), arr(c_i32, offsets),
        len(k.instrs), arr(c_i32, words), arr(ctypes.c_double, lits), k.result,
        ctypes.c_void_p(out.ctypes.data), arr(c_i64, [int(np.prod(shapes[op.tensor], dtype=np.int64))
                                                      for op in list(k.reads) + [k.write]]))
    if rc != 0:
        raise RuntimeError(f"ref_interp_kernel2 failed ({rc})")


def run_kernel(k, bounds, vals, shapes, tensors, epoch=0, f64=False, abs_into=None):
    """f64: True = the float64 shadow of the kernel (oracle/refinterp_body.h) on float64 tensors; "c64" = the kernel of a
    compile[float64] program (float64 arithmetic AND float64 constants: the reference's own path for T = float64).
    abs_into: instead of the written tensor, add the MAGNITUDE of every term to this array (affine kernels only)."""
    if k.index_instrs:
        if abs_into is not None:
            raise NotImplementedError("term magnitudes of a kernel with computed indices")
        return _run_kernel_indexed(k, bounds, vals, shapes, tensors, epoch, f64)
    lib = refcpu.lib()
    interp = {False: lib.ref_interp_kernel, True: lib.ref_interp_kernel_f64, "c64": lib.ref_interp_kernel_c64}[f64]
    interp.restype = ctypes.c_int
    nl = len(k.loops)
    loop_index = {lp.reg: i for i, lp in enumerate(k.loops)}
    typ = infer_types(k, vals)

    def affine(op):
        shp = shapes[op.tensor]
        out = [0] * (1 + nl)
        if op.raw:
            strides = [1]
        else:
            strides = [1] * len(shp)
            for d in range(len(shp) - 2, -1, -1):
                strides[d] = strides[d + 1] * shp[d + 1]
        for d, lin in enumerate(op.dims):
            out[0] += strides[d] * lin.constant
            for r, f in lin.factors.items():
                if r in loop_index:
                    out[1 + loop_index[r]] += strides[d] * f
                else:
                    out[0] += strides[d] * f * vals[r]
        return out

    instr_words, lits = [], []
    for ins in k.instrs:
        kd = ins.kind
        lit = 0.0
        if kd in ("shape", "len", "shapelen", "epoch"):
            # host-evaluated builtins (model.nim:83-104) become index literals
            if kd == "shape":
                lit = shapes[ins.extra[0]][ins.extra[1]]
            elif kd == "len":
                lit = int(np.prod(shapes[ins.extra[0]], dtype=np.int64))
            elif kd == "shapelen":
                lit = len(shapes[ins.extra[0]])
            else:
                lit = epoch
            code = 1
        elif kd in ("scalar", "index", "boolean"):
            code = _OPC[kd]
            lit = float(ins.extra)
        elif (kd, typ[ins.args[0]] if ins.args else None) in _OPC:
            code = _OPC[(kd, typ[ins.args[0]])]
        else:
            code = _OPC[kd]
        a = ins.args + [0, 0, 0]
        instr_words += [code, ins.res, a[0], a[1], a[2]]
        lits.append(lit)

    ninstr = len(k.instrs)
    c_i64, c_i32 = ctypes.c_int64, ctypes.c_int32
    starts = (c_i64 * max(nl, 1))(*[bounds[lp.reg][0] for lp in k.loops])
    stops = (c_i64 * max(nl, 1))(*[bounds[lp.reg][1] for lp in k.loops])
    lregs = (c_i32 * max(nl, 1))(*[lp.reg for lp in k.loops])
    nreads = len(k.reads)
    rptrs = (ctypes.c_void_p * max(nreads, 1))(*[tensors[r.tensor].ctypes.data for r in k.reads])
    rregs = (c_i32 * max(nreads, 1))(*[r.reg for r in k.reads])
    raff = []
    for r in k.reads:
        raff += affine(r)
    raff_c = (c_i64 * max(len(raff), 1))(*raff)
    instr_c = (c_i32 * max(len(instr_words), 1))(*instr_words)
    lits_c = (ctypes.c_double * max(ninstr, 1))(*lits)
    waff = (c_i64 * (1 + nl))(*affine(k.write))
    out = tensors[k.write.tensor] if abs_into is None else abs_into
    rc = interp(nl, starts, stops, lregs, k.nregs + 1, nreads, rptrs, rregs, raff_c, ninstr, instr_c,
                lits_c, k.result, ctypes.c_void_p(out.ctypes.data), waff, 0 if abs_into is None else 2)
    if rc != 0:
        raise RuntimeError(f"ref_interp_kernel failed ({rc})")


def contraction_pattern(k):
    """If k is `C[i,j] += A(i,k) * B(k,j)` return (a_op, b_op, trans_a, trans_b, regs) — lets
    large cases run on ref_sgemm, whose loop nest is the same summation (tests check equality)."""
    if len(k.instrs) != 1 or k.instrs[0].kind != "mul" or len(k.reads) != 2 or k.write.raw or len(k.write.dims) != 2:
        return None
    wi, wj = k.write.dims[0].only_register(), k.write.dims[1].only_register()
    if not wi or not wj or wi == wj or len(k.loops) != 3:
        return None
    kk = [lp.reg for lp in k.loops if lp.reg not in (wi, wj)]
    if len(kk) != 1:
        return None
    kk = kk[0]
    if sorted(k.instrs[0].args) != sorted(r.reg for r in k.reads) or k.result != k.instrs[0].res:
        return None

    def regs_of(op):
        if op.raw or len(op.dims) != 2:
            return None
        a, b = op.dims[0].only_register(), op.dims[1].only_register()
        return (a, b) if a and b else None

    r0, r1 = regs_of(k.reads[0]), regs_of(k.reads[1])
    if r0 is None or r1 is None:
        return None
    for a_op, ra, b_op, rb in ((k.reads[0], r0, k.reads[1], r1), (k.reads[1], r1, k.reads[0], r0)):
        if set(ra) == {wi, kk} and set(rb) == {kk, wj}:
            return a_op, b_op, ra == (kk, wi), rb == (wj, kk)
    return None


class Model:
    """CPU-path Model: params persist and are updated in place by optimizer kernels
    (model.nim:284), result tensors are zeroed before every call (model.nim:295-300)."""

    def __init__(self, text, fast_contractions=True, threads=1, shadow=False):
        """shadow=True: the float64 SHADOW of the program — the same kernel list (derive, dead-kernel
        elimination, loop order) evaluated in float64 on float64 tensors, i.e. the exact value of what
        the reference computes in float32.  Tests hold the GPU to 1e-5 of it and the float32 oracle to
        its own sequential-summation bound, instead of widening a GPU-vs-oracle tolerance."""
        self.prog = parse(text)
        # compile[float64] (model.nim:253-260; header `kd 1 f64`): the reference instantiates every kernel over float64 —
        # double tensors, double constants, libm's double functions, the same loop nests (refinterp.c "_c64").
        self.c64 = getattr(self.prog, "scalar", "f32") == "f64"
        self.dtype = np.float64 if (shadow or self.c64) else np.float32
        self.compiled = {name: compile_target(self.prog, name) for name in list(self.prog.targets)}
        self.params = {}
        self.epoch = 0
        self.fast = fast_contractions
        self.threads = threads
        self.last = {}
        self.caches = {}
        # TensorRandom (`rand`, dropout's mask): the reference refills them from Nim's global RNG on
        # every call (model.nim:286-294) — unpinned.  Here: numpy's generator, or, for parity tests,
        # the numbers another implementation drew (random_override[tensor id]).
        self.rng = np.random.default_rng(0)
        self.random_override = {}
        # shadow only: for the tensors in track_abs, abs_terms[t] = sum of the magnitudes of the terms every kernel adds
        # into t (same shape as t) — what a float32 summation error of t is proportional to (tests/parity.py)
        self.track_abs = set()
        self.abs_terms = {}
        for tid, t in self.prog.tensors.items():
            if t["kind"] == "param":
                self.params[tid] = np.zeros(t["shape"], dtype=self.dtype)
            elif t["kind"] == "cache":      # model.nim:248-249: zero tensors that persist across calls
                self.caches[tid] = np.zeros(t["shape"], dtype=self.dtype)

    def kernel_count(self, target):
        return len(self.compiled[target][1])

    def call(self, target, inputs):
        self._forward_backward(target, inputs, 1.0, stop_at_update=False)
        output = self.compiled[target][0]
        return self.last[output] if output else None

    def apply(self, target, inputs):
        self.call(target, inputs)

    def fit(self, target, inputs, batch_size=32):
        """Model.fit, model.nim:413-454: epoch += 1 ONCE (436), then the target once per mini-batch of `batch_size`
        leading rows of every input (batchCount = rows of the FIRST input div batchSize, 425: the tail is dropped;
        viewFirst slices, 441-443); result tensors start from zero for every batch (447-449: here every call does)."""
        inputs = list(inputs.items()) if isinstance(inputs, dict) else list(inputs)
        if not inputs:      # model.nim:418-419
            raise RuntimeError("Model.fit requires at least one input tensor. Use Model.apply instead if the target has zero inputs.")
        if target not in self.prog.targets:     # model.nim:420-421
            raise RuntimeError(target + " is not a target of the model")
        for name, _ in inputs:                  # model.nim:429-430
            if name not in self.prog.inputs:
                raise RuntimeError(name + " is not an input to the model")
        batches = int(np.asarray(inputs[0][1]).shape[0]) // batch_size
        self.epoch += 1
        for b in range(batches):
            lo = b * batch_size
            self.apply(target, {name: np.asarray(arr)[lo:lo + batch_size] for name, arr in inputs})

    # ---- split step for the data-parallel tests: [forward + backward] | all-reduce | [update] ----
    def param_grads(self, target):
        """[(param tensor id, gradient tensor id)] of the target's optimizer (GenGradient markers)."""
        return list(self.prog.param_grads[target])

    def run_backward(self, target, inputs, grad_scale=1.0):
        self._forward_backward(target, inputs, grad_scale, stop_at_update=True)

    def run_update(self, target):
        for k in self._pending:
            self._run_one(k, *self._pending_state)
        self._pending = []

    def _run_one(self, k, infos, shapes, tensors, grad_scale):
        bounds, vals = infos[id(k)]
        wt = k.write.tensor
        shadow = "c64" if self.c64 else self.dtype == np.float64
        if wt not in tensors:
            tensors[wt] = np.zeros(shapes[wt], dtype=self.dtype)
        pat = contraction_pattern(k) if self.fast else None
        if k.is_seed:
            # gradLoss{i} = 1 (passes.nim:594-596), times B_local/B_global under data parallelism
            tensors[wt] += self.dtype(np.float32(grad_scale))
        elif pat is not None and self.c64:
            a_op, b_op, ta, tb = pat
            refcpu.dgemm64(tensors[a_op.tensor], tensors[b_op.tensor], ta, tb, out=tensors[wt])
        elif pat is not None and shadow:
            a_op, b_op, ta, tb = pat
            a, b = tensors[a_op.tensor], tensors[b_op.tensor]
            tensors[wt] += (a.T if ta else a) @ (b.T if tb else b)
        elif pat is not None:
            a_op, b_op, ta, tb = pat
            refcpu.sgemm(tensors[a_op.tensor], tensors[b_op.tensor], ta, tb, out=tensors[wt], threads=self.threads)
        else:
            run_kernel(k, bounds, vals, shapes, tensors, self.epoch, f64=shadow)
        if shadow and wt in self.track_abs:
            acc = self.abs_terms.setdefault(wt, np.zeros(shapes[wt], dtype=np.float64))
            if acc is None:
                pass
            elif k.is_seed:
                acc += abs(float(np.float32(grad_scale)))
            elif pat is not None:
                a_op, b_op, ta, tb = pat
                a, b = np.abs(tensors[a_op.tensor]), np.abs(tensors[b_op.tensor])
                acc += (a.T if ta else a) @ (b.T if tb else b)
            else:
                try:
                    run_kernel(k, bounds, vals, shapes, tensors, self.epoch, f64=True, abs_into=acc)
                except NotImplementedError:
                    self.abs_terms[wt] = None       # (a writer with computed indices: no magnitude sum for this tensor)

    def _forward_backward(self, target, inputs, grad_scale, stop_at_update):
        if target not in self.compiled:
            raise KeyError(target + " is not a target of the model")      # model.nim:395-396
        output, kernels, all_kernels = self.compiled[target]
        shapes, tensors = {}, {}
        for name, arr in inputs.items():
            if name not in self.prog.inputs:
                raise KeyError(name + " is not an input to the model")    # model.nim:358-359
            tid = self.prog.inputs[name]
            if not self.c64:
                arr = np.ascontiguousarray(arr, dtype=np.float32)
            arr = np.ascontiguousarray(arr, dtype=self.dtype)
            static = self.prog.tensors[tid]["shape"]
            if static:
                if len(static) != arr.ndim or any(s >= 0 and s != a for s, a in zip(static, arr.shape)):
                    raise ShapeError(f"input {name}: expected shape {static}, got {list(arr.shape)}")
            shapes[tid], tensors[tid] = list(arr.shape), arr
        for tid, p in list(self.params.items()) + list(self.caches.items()):
            shapes[tid], tensors[tid] = list(p.shape), p
        live = {id(k) for k in kernels}
        infos = {}
        for k in all_kernels:
            for r in k.reads:                  # random tensors: shaped like their source, drawn per call
                t = self.prog.tensors[r.tensor]
                src = self.prog.shape_copy.get(r.tensor)
                if t["kind"] == "random" and r.tensor not in shapes and src in shapes:
                    shapes[r.tensor] = list(shapes[src])
                    if r.tensor in self.random_override:
                        tensors[r.tensor] = np.ascontiguousarray(self.random_override[r.tensor], dtype=self.dtype)
                    else:
                        lo, hi = t["range"]
                        tensors[r.tensor] = (lo + (hi - lo) * self.rng.random(shapes[r.tensor], dtype=np.float32)).astype(self.dtype)
            if any(r.tensor not in shapes for r in k.reads):
                if id(k) in live:
                    missing = [r.tensor for r in k.reads if r.tensor not in shapes]
                    raise ShapeError(f"tensors {missing} are read before their shape is known (missing input?)")
                continue
            try:
                infos[id(k)] = infer_kernel(self.prog, k, shapes, self.epoch)
            except (ShapeError, KeyError):
                if id(k) in live:
                    raise
        first_update = len(kernels)
        for i, k in enumerate(kernels):
            if self.prog.tensors[k.write.tensor]["kind"] in ("param", "cache"):
                first_update = i
                break
        stop = first_update if stop_at_update else len(kernels)
        self.abs_terms = {}
        for k in kernels[:stop]:
            self._run_one(k, infos, shapes, tensors, grad_scale)
        # gradient tensors of parameters must exist even if nothing wrote them
        for _, gt in self.prog.param_grads.get(target, []):
            if gt not in tensors and gt in shapes:
                tensors[gt] = np.zeros(shapes[gt], dtype=self.dtype)
        self._pending = list(kernels[stop:])
        self._pending_state = (infos, shapes, tensors, grad_scale)
        self.last = tensors
        self.last_shapes = shapes
