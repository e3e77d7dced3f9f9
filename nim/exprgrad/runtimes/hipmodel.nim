# exprgrad/runtimes/hipmodel.nim — Level 2 of the integration (INTEGRATION.md): hand a GPU target's kernel list to
# libexprgrad_hip.so (group 3 of include/exprgrad_hip.h) instead of lowering it to InstrGpu bodies + OpenCL text.
#
# Written against can-lehmann/exprgrad v1 (ir.nim, parser.nim, passes.nim, model.nim, tensors.nim).  No Nim compiler
# exists in the build image of this repository, so this file has never been compiled; what it EMITS is pinned
# instead: tests/golden/handwritten/xor_from_scratch.kd is the text `toKd` must produce for
# examples/xor_from_scratch/xor_from_scratch.nim:19-31, written by hand following the register numbering of
# parser.nim:159-259, and it is compiled and run on the GPU (tests/test_gpu_handwritten.py, tests/cabi_harness.c).
#
# Contents:
#   toKd(program)        Program (after foldLinearIndices + deduplicateReads, before generate) -> kernel-description text
#   HipModel / newHipModel / call / apply / fit / param access: what model.nim does for a CompileGpu target
#     (model.nim:215-251, 302-318, 357-383, 392-454), over eg_model_*.
# nim/PATCHES.md lists the three places of the reference that select this module.

import std/[tables, sets, strutils, algorithm]
import ../ir, ../passes, ../tensors
import hip

type
  EgModel {.importc: "eg_model", header: "exprgrad_hip.h", incompleteStruct.} = object

{.push importc, cdecl, header: "exprgrad_hip.h".}
proc eg_model_compile(ctx: pointer, text: cstring, res: ptr ptr EgModel): cint
proc eg_model_free(model: ptr EgModel): cint
proc eg_model_set_input_host(model: ptr EgModel, name: cstring, host: ptr float32, rank: cint, shape: ptr int64): cint
proc eg_model_clear_inputs(model: ptr EgModel): cint
proc eg_model_run(model: ptr EgModel, target: cstring): cint
proc eg_model_fit(model: ptr EgModel, target: cstring, nInputs: cint, names: ptr cstring, data: ptr ptr float32,
                  onDevice: ptr cint, ranks: ptr cint, shapes8: ptr int64, batchSize: int64): cint
proc eg_model_output_shape(model: ptr EgModel, target: cstring, rank: ptr cint, shape8: ptr int64): cint
proc eg_model_read_output(model: ptr EgModel, target: cstring, host: ptr float32, count: int64): cint
proc eg_model_param_write(model: ptr EgModel, tensor: cint, host: ptr float32, count: int64): cint
proc eg_model_param_read(model: ptr EgModel, tensor: cint, host: ptr float32, count: int64): cint
proc eg_model_set_epoch(model: ptr EgModel, epoch: int64): cint
proc eg_model_epoch(model: ptr EgModel): int64
proc eg_model_keep_values(model: ptr EgModel, on: cint): cint
proc eg_model_plan_text(model: ptr EgModel): cstring
proc eg_model_launch_text(model: ptr EgModel, target: cstring): cstring
proc eg_model_state_bytes(model: ptr EgModel, bytes: ptr csize_t): cint
proc eg_model_store_state(model: ptr EgModel, buf: pointer, cap: csize_t, written: ptr csize_t): cint
proc eg_model_load_state(model: ptr EgModel, buf: pointer, bytes: csize_t, consumed: ptr csize_t): cint
proc eg_last_error(): cstring
# compile[float64] (model.nim:253-260): the float64 twins of the typed entry points (include/exprgrad_hip.h)
proc eg_model_set_input_host_f64(model: ptr EgModel, name: cstring, host: ptr float64, rank: cint, shape: ptr int64): cint
proc eg_model_fit_f64(model: ptr EgModel, target: cstring, nInputs: cint, names: ptr cstring, data: ptr ptr float64,
                      onDevice: ptr cint, ranks: ptr cint, shapes8: ptr int64, batchSize: int64): cint
proc eg_model_read_output_f64(model: ptr EgModel, target: cstring, host: ptr float64, count: int64): cint
proc eg_model_param_write_f64(model: ptr EgModel, tensor: cint, host: ptr float64, count: int64): cint
proc eg_model_param_read_f64(model: ptr EgModel, tensor: cint, host: ptr float64, count: int64): cint
{.pop.}

# The typed calls by the model's T (float32 / float64: toScalarType, model.nim:253-260)
template setInputHost[T](model: ptr EgModel, name: cstring, host: ptr T, rank: cint, shape: ptr int64): cint =
  when T is float64: eg_model_set_input_host_f64(model, name, host, rank, shape)
  else: eg_model_set_input_host(model, name, host, rank, shape)
template readOutput[T](model: ptr EgModel, target: cstring, host: ptr T, count: int64): cint =
  when T is float64: eg_model_read_output_f64(model, target, host, count)
  else: eg_model_read_output(model, target, host, count)
template paramWrite[T](model: ptr EgModel, tensor: cint, host: ptr T, count: int64): cint =
  when T is float64: eg_model_param_write_f64(model, tensor, host, count)
  else: eg_model_param_write(model, tensor, host, count)
template paramRead[T](model: ptr EgModel, tensor: cint, host: ptr T, count: int64): cint =
  when T is float64: eg_model_param_read_f64(model, tensor, host, count)
  else: eg_model_param_read(model, tensor, host, count)

# Status codes of include/exprgrad_hip.h that map onto exceptions of their own (tests/test_errors.nim expects them).
const
  EG_ERR_RUNTIME = 6
  EG_ERR_SHAPE = 7

proc check(status: cint) =
  if status != 0:
    let msg = $eg_last_error()
    case int(status):
      of EG_ERR_RUNTIME: raise RuntimeError(msg: msg)
      of EG_ERR_SHAPE: raise ShapeError(msg: msg)
      else: raise GpuError(msg: msg)

# ------------------------------------------------------------------------------------------------------------------
# toKd: Program -> kernel-description text (grammar: DESIGN.md "Kernel-description text")
# ------------------------------------------------------------------------------------------------------------------

type
  KdEmitter = object
    lines: seq[string]
    warnedSchedule: bool

  # Array-typed registers (InstrArray / InstrArrayRead on an array of arrays, ir.nim:69) have no counterpart in the
  # text: they are tracked symbolically and every scalar `arr[i]` becomes the select chain
  #   select(i == 0, a0, select(i == 1, a1, ... a(n-1)))
  # over the same element registers (identical value for every in-range index; what exprgrad_amd/dsl.py emits).
  ArrayValue = ref object
    elems: seq[RegId]          # element registers (scalars, or registers that hold arrays themselves)
    isView: bool               # arr[i] of an array of arrays: one row, chosen at run time by `index`
    rows: seq[ArrayValue]
    index: RegId

proc kdFloat(value: float64): string =
  ## Literal in a form strtod() reads back exactly (the library parses "ins scalar" with strtod).
  if value != value: "nan"
  elif value == Inf: "inf"
  elif value == NegInf: "-inf"
  else: formatFloat(value, ffDefault, 17)

proc kdLinear(index: LinearIndex): string =
  ## "L <constant> <n> (<reg> <factor>)*", factors in register order so that the text is deterministic
  var regs: seq[int] = @[]
  for reg, factor in index.factors:
    if factor != 0:
      regs.add(int(reg))
  regs.sort()
  result = "L " & $index.constant & " " & $regs.len
  for reg in regs:
    result &= " " & $reg & " " & $index.factors[RegId(reg)]

proc kdName(kind: InstrKind): string =
  result = ($kind)[len("Instr")..^1].toLowerAscii()

proc kdOp(op: TensorOp, keyword: string, tensorIds: Table[TensorId, int]): string =
  ## "read|write <tensor> <data reg> <raw 0|1> <ndims> <L...>"; tensorIds maps the gradient placeholders of a custom
  ## gradient (TensorId(-1), TensorId(-2), ... handed out by parser.nim:142-146) to -<tensor they are the gradient of>
  var tensor = int(op.tensor)
  if op.tensor in tensorIds:
    tensor = tensorIds[op.tensor]
  result = keyword & " " & $tensor & " " & $int(op.data) & " " & $ord(op.isRaw) & " " & $op.dims.len
  for dim in op.dims:
    result &= " " & dim.kdLinear()

type InstrSink = enum SinkSetup, SinkIdx, SinkIns

const NO_ARGS: seq[RegId] = @[]

proc emitInstr(em: var KdEmitter, sink: InstrSink, kind: string, res: RegId, args: openArray[RegId], extra: string = "") =
  const KEYWORD: array[InstrSink, string] = ["setup", "idx", "ins"]
  var line = KEYWORD[sink] & " " & kind & " " & $int(res) & " " & $args.len
  for arg in args:
    line &= " " & $int(arg)
  if extra.len > 0:
    line &= " " & extra
  em.lines.add("  " & line)

proc emitScalarInstr(em: var KdEmitter, sink: InstrSink, instr: Instr, tensorIds: Table[TensorId, int]) =
  ## One non-array instruction of ir.nim:51-76 in the text form of csrc/host/kd.cpp:parse_instr
  var extra = ""
  case instr.kind:
    of InstrIndex: extra = $instr.indexLit
    of InstrScalar: extra = kdFloat(instr.scalarLit)
    of InstrBoolean: extra = $ord(instr.booleanLit)
    of InstrShape, InstrLen, InstrShapeLen:
      var tensor = int(instr.tensor)
      if instr.tensor in tensorIds:
        tensor = tensorIds[instr.tensor]
      extra = $tensor
      if instr.kind == InstrShape:
        extra &= " " & $instr.dim
    of InstrAdd, InstrSub, InstrMul, InstrDiv, InstrIndexDiv, InstrMod, InstrWrap,
       InstrNegate, InstrSin, InstrCos, InstrExp, InstrPow, InstrSqrt,
       InstrLog, InstrLog10, InstrLog2, InstrLn,
       InstrEq, InstrLt, InstrLe, InstrAnd, InstrOr, InstrSelect,
       InstrToScalar, InstrToIndex, InstrEpoch:
      discard
    else:
      # InstrRead / InstrWrite / InstrOverwrite, loops, threads, GPU instructions: products of passes that run after
      # generate (inlineTensorOps, inlineLoops, ...); a source Program (model.nim:232-236) holds none of them
      raise GeneratorError(msg: "toKd: " & $instr.kind & " cannot appear in a kernel description (call toKd on the source Program)")
  em.emitInstr(sink, instr.kind.kdName(), instr.res, instr.args, extra)

proc readArray(em: var KdEmitter, sink: InstrSink, value: ArrayValue, index: RegId,
               arrays: var Table[RegId, ArrayValue], nextReg: var int): RegId

proc selectChain(em: var KdEmitter, sink: InstrSink, index: RegId, choices: seq[RegId], nextReg: var int): RegId =
  ## select(index == 0, c0, select(index == 1, c1, ... c(n-1))) with fresh registers; returns the result register
  result = choices[^1]
  for it in countdown(choices.len - 2, 0):
    let
      lit = RegId(nextReg + 1)
      cond = RegId(nextReg + 2)
      sel = RegId(nextReg + 3)
    nextReg += 3
    em.emitInstr(sink, "index", lit, NO_ARGS, $it)
    em.emitInstr(sink, "eq", cond, [index, lit])
    em.emitInstr(sink, "select", sel, [cond, choices[it], result])
    result = sel

proc readArray(em: var KdEmitter, sink: InstrSink, value: ArrayValue, index: RegId,
               arrays: var Table[RegId, ArrayValue], nextReg: var int): RegId =
  ## The scalar `value[index]`; for a row view of an array of arrays: select over the rows of `row[index]`
  if value.isView:
    var perRow: seq[RegId] = @[]
    for row in value.rows:
      perRow.add(em.readArray(sink, row, index, arrays, nextReg))
    result = em.selectChain(sink, value.index, perRow, nextReg)
  else:
    result = em.selectChain(sink, index, value.elems, nextReg)

proc emitInstrs(em: var KdEmitter, sink: InstrSink, instrs: seq[Instr], tensorIds: Table[TensorId, int],
                arrays: var Table[RegId, ArrayValue], subs: var Table[RegId, RegId], nextReg: var int) =
  ## `subs`: result registers of array reads are replaced by the register their select chain ends in
  for original in instrs:
    var instr = original
    for arg in instr.args.mitems:
      if arg in subs:
        arg = subs[arg]
    case instr.kind:
      of InstrArray:
        arrays[instr.res] = ArrayValue(elems: instr.args)
      of InstrArrayLen:
        em.emitInstr(sink, "index", instr.res, NO_ARGS, $arrays[instr.args[0]].elems.len)
      of InstrArrayRead:
        let value = arrays[instr.args[0]]
        var nested = false
        if not value.isView and value.elems.len > 0 and value.elems[0] in arrays:
          nested = true
        if nested:                                 # arr[y] of an array of arrays: a row that waits for the next index
          var view = ArrayValue(isView: true, index: instr.args[1])
          for elem in value.elems:
            view.rows.add(arrays[elem])
          arrays[instr.res] = view
        else:
          subs[instr.res] = em.readArray(sink, value, instr.args[1], arrays, nextReg)
      else:
        em.emitScalarInstr(sink, instr, tensorIds)

proc dependsOnIterators(instrs: seq[Instr], loopRegs: HashSet[RegId]): HashSet[RegId] =
  ## Result registers of `instrs` whose value changes inside the loop nest
  result = loopRegs
  for instr in instrs:
    for arg in instr.args:
      if arg in result:
        result.incl(instr.res)

proc warnSchedule(em: var KdEmitter, what: string) =
  if not em.warnedSchedule:
    em.warnedSchedule = true
    stderr.writeLine("exprgrad/hip: schedule directives (" & what & ") are ignored by the HIP backend: contractions, " &
                     "convolutions and reductions run on its own kernels whatever the schedule says")

proc emitKernel(em: var KdEmitter, kernel: Kernel, tensorIds: Table[TensorId, int]) =
  if kernel.conds.len > 0:
    raise GeneratorError(msg: "toKd: kernel conditions are produced by later passes; call toKd on the source Program")
  var
    nextReg = kernel.regs.len
    arrays = initTable[RegId, ArrayValue]()
    subs = initTable[RegId, RegId]()
    loopRegs = initHashSet[RegId]()
    body: KdEmitter                     # the kernel's lines are collected first: "kernel <nregs>" needs the final count
  for loop in kernel.loops:
    loopRegs.incl(loop.iter)
    if loop.schedule != DEFAULT_LOOP_SCHEDULE:
      em.warnSchedule("tile / tileSize / parallel / shareCache")
  for read in kernel.reads:
    if read.schedule != DEFAULT_TENSOR_SCHEDULE:
      em.warnSchedule("cache")

  # host-evaluated instructions: Kernel.setup and the setup of explicit loop bounds (shape() / len() / literals)
  body.emitInstrs(SinkSetup, kernel.setup, tensorIds, arrays, subs, nextReg)
  for loop in kernel.loops:
    if loop.hasBounds:
      body.emitInstrs(SinkSetup, loop.start.setup, tensorIds, arrays, subs, nextReg)
      body.emitInstrs(SinkSetup, loop.stop.setup, tensorIds, arrays, subs, nextReg)
  # LinearIndex.setup of the operands (ir.nim:120-123): what depends on an iterator is computed per point of the loop
  # nest ("idx", e.g. maxpool2's `y div 2`, dnn.nim:59-71), the rest on the host ("setup")
  var operandSetup: seq[Instr] = @[]
  for read in kernel.reads:
    for dim in read.dims:
      operandSetup.add(dim.setup)
  for dim in kernel.write.dims:
    operandSetup.add(dim.setup)
  let varying = operandSetup.dependsOnIterators(loopRegs)
  var hostPart, loopPart: seq[Instr]
  for instr in operandSetup:
    if instr.res in varying: loopPart.add(instr)
    else: hostPart.add(instr)
  body.emitInstrs(SinkSetup, hostPart, tensorIds, arrays, subs, nextReg)

  for loop in kernel.loops:
    var name = kernel.regs[loop.iter].name
    if name.len == 0:
      name = "i" & $int(loop.iter)
    name = name.replace(" ", "_")
    if loop.hasBounds:
      if loop.step != 1 and loop.step != 0:
        raise GeneratorError(msg: "toKd: loop step " & $loop.step & " (only unit steps exist before tileLoops)")
      body.lines.add("  loop " & $int(loop.iter) & " " & name & " 1 " & loop.start.kdLinear() & " " & loop.stop.kdLinear())
    else:
      body.lines.add("  loop " & $int(loop.iter) & " " & name & " 0")

  body.emitInstrs(SinkIdx, loopPart, tensorIds, arrays, subs, nextReg)
  for read in kernel.reads:
    body.lines.add("  " & read.kdOp("read", tensorIds))
  body.emitInstrs(SinkIns, kernel.expr.instrs, tensorIds, arrays, subs, nextReg)
  var res = kernel.expr.res
  if res in subs:
    res = subs[res]
  body.lines.add("  result " & $int(res))
  var write = kernel.write
  write.data = res
  body.lines.add("  " & write.kdOp("write", tensorIds))

  em.lines.add("  kernel " & $nextReg)
  em.lines.add(body.lines)
  if kernel.grad.isCustom:
    # KernelGradient (ir.nim:203-209): kernels the user wrote in a customGrad block; the placeholder TensorId(-k) of
    # grad(t) (parser.nim:142-146, kernel.grad.tensors: t -> placeholder) becomes -t, which is how the text names
    # "the gradient tensor of t"
    var gradIds = tensorIds
    for tensor, placeholder in kernel.grad.tensors:
      var target = tensor
      if tensor in kernel.grad.subs:
        target = kernel.grad.subs[tensor]
      gradIds[placeholder] = -int(target)
    em.lines.add("  customgrad")
    for gradKernel in kernel.grad.kernels:
      em.emitKernel(gradKernel, gradIds)
    em.lines.add("  endcustomgrad")
  em.lines.add("  endkernel")

proc emitReshape(em: var KdEmitter, kernel: Kernel, shapeLines: var seq[string], shaped: var HashSet[TensorId]) =
  ## GenReshape (ir.nim:196-201) as generate would expand it (passes.nim:643-688): a raw copy over len(source) plus a
  ## ShapeDims constraint whose -1 entry is len(source) div (product of the others)
  let
    src = int(kernel.generator.tensor)
    dest = int(kernel.write.tensor)
  em.lines.add("  kernel 3")
  em.lines.add("  setup len 3 0 " & $src)
  em.lines.add("  loop 2 reshape.it 1 L 0 0 L 0 1 3 1")
  em.lines.add("  read " & $src & " 1 1 1 L 0 1 2 1")
  em.lines.add("  result 1")
  em.lines.add("  write " & $dest & " 1 1 1 L 0 1 2 1")
  em.lines.add("  endkernel")
  if kernel.write.tensor notin shaped:
    shaped.incl(kernel.write.tensor)
    var
      prod = 1
      dims = "shapedims " & $dest & " " & $kernel.generator.reshape.len
      setup: seq[string] = @[]
    for size in kernel.generator.reshape:
      if size >= 0:
        prod *= size
    for size in kernel.generator.reshape:
      if size >= 0:
        dims &= " L " & $size & " 0"
      else:
        setup.add("shapesetup " & $dest & " len 1 0 " & $src)
        setup.add("shapesetup " & $dest & " index 2 0 " & $prod)
        setup.add("shapesetup " & $dest & " indexdiv 3 2 1 2")
        dims &= " L 0 1 3 1"
    shapeLines.add(dims)
    shapeLines.add(setup)

proc emitShapeConstraint(constr: ShapeConstraint, shapeLines: var seq[string], shaped: var HashSet[TensorId]) =
  ## User constraints of a target (parser.nim:340-357, 378-383).  A ShapeDims entry was built with a fresh register
  ## file per dimension (parser.nim:349-353): registers are renumbered so that the dimensions of one constraint can
  ## share the "shapesetup" instruction list.
  if constr.dest in shaped:
    return
  case constr.kind:
    of ShapeCopy:
      shaped.incl(constr.dest)
      shapeLines.add("shapecopy " & $int(constr.dest) & " " & $int(constr.src))
    of ShapeDims:
      shaped.incl(constr.dest)
      var
        offset = 0
        line = "shapedims " & $int(constr.dest) & " " & $constr.dims.len
        setup: seq[string] = @[]
      for dim in constr.dims:
        var highest = 0
        for instr in dim.setup:
          var text = "shapesetup " & $int(constr.dest) & " " & instr.kind.kdName() & " " & $(int(instr.res) + offset) & " " & $instr.args.len
          for arg in instr.args:
            text &= " " & $(int(arg) + offset)
          case instr.kind:
            of InstrIndex: text &= " " & $instr.indexLit
            of InstrShape: text &= " " & $int(instr.tensor) & " " & $instr.dim
            of InstrLen, InstrShapeLen: text &= " " & $int(instr.tensor)
            of InstrAdd, InstrSub, InstrMul, InstrNegate, InstrIndexDiv, InstrMod, InstrEpoch: discard
            else: raise GeneratorError(msg: "toKd: " & $instr.kind & " in a shape constraint (host-evaluated: index arithmetic, shape(), len())")
          setup.add(text)
          highest = max(highest, int(instr.res))
        var shifted = LinearIndex(constant: dim.constant)
        for reg, factor in dim.factors:
          shifted.factors[RegId(int(reg) + offset)] = factor
          highest = max(highest, int(reg))
        line &= " " & shifted.kdLinear()
        offset += highest
      shapeLines.add(line)
      shapeLines.add(setup)
    else:
      discard     # ShapeLinear / ShapeRank are inferred (inferShapeConstraints), never written by the user

proc toKd*(source: Program): string =
  ## Kernel-description text of a Program as `toProgram` built it (parser.nim:404-417), i.e. model.source
  ## (model.nim:232-236).  The passes that only normalise — makeTensorLookups, deadCodeElim, foldLinearIndices,
  ## deduplicateReads (model.nim:47-50) — run here on a clone; generate and everything after it is the library's job.
  # the scalar type (Scalar32 / Scalar64, ir.nim; set by compile[T] through toScalarType, model.nim:253-260) is the header's
  # third word: a float64 program runs on eg_dgemm + generated kernels over double
  let program = source.clone()
  program.makeTensorLookups()
  program.deadCodeElim()
  program.foldLinearIndices()
  program.deduplicateReads()

  var
    em: KdEmitter
    shapeLines: seq[string] = @[]
    shaped = initHashSet[TensorId]()
    targetLines: seq[string] = @[]
  let noIds = initTable[TensorId, int]()

  # targets in name order: Table iteration order depends on the hash seed, the text should not
  var names: seq[string] = @[]
  for name in program.targets.keys:
    names.add(name)
  names.sort()
  for name in names:
    let target = program.targets[name]
    if name.len == 0 or name.contains({' ', '\t', '\n'}):
      raise GeneratorError(msg: "toKd: target name \"" & name & "\" is empty or contains whitespace (the text is token based)")
    em.lines = @[]
    em.lines.add("target " & name & " " & $int(target.output))
    for kernel in target.kernels:
      case kernel.generator.kind:
        of GenNone: em.emitKernel(kernel, noIds)
        of GenBackwards: em.lines.add("  backwards " & $int(kernel.generator.tensor))
        of GenGradient: em.lines.add("  gradient " & $int(kernel.generator.tensor) & " " & $int(kernel.write.tensor))
        of GenReshape: em.emitReshape(kernel, shapeLines, shaped)
    em.lines.add("endtarget")
    targetLines.add(em.lines)
    for constr in target.shapes:
      constr.emitShapeConstraint(shapeLines, shaped)

  var lines = @[if source.scalarType == Scalar64: "kd 1 f64" else: "kd 1 f32"]
  for it, def in program.tensors:
    # tensor <id> input|param|result|cache|random <name|-> <rank|-1> <dims...> [<lo> <hi>]
    var
      name = def.name.replace(" ", "_")
      line = "tensor " & $(it + 1) & " "
    if name.len == 0:
      name = "-"
    case def.kind:
      of TensorInput:
        line &= "input " & name
        if def.shape.len == 0: line &= " -1"          # input("x") without a static shape (parser.nim:724-731)
        else: line &= " " & $def.shape.len & " " & def.shape.join(" ")
      of TensorParam:
        line &= "param " & name & " " & $def.shape.len
        for size in def.shape: line &= " " & $size
        line &= " " & kdFloat(def.initRange.a) & " " & kdFloat(def.initRange.b)
      of TensorCache:
        # shaped like the tensor it shadows (parser.nim:296-303: a parameter)
        let shape = program.tensors[def.cache].shape
        line &= "cache " & name & " " & $shape.len
        for size in shape: line &= " " & $size
      of TensorRandom:
        line &= "random " & name & " -1 " & kdFloat(def.randomRange.a) & " " & kdFloat(def.randomRange.b)
      of TensorResult:
        line &= "result " & name & " -1"
    lines.add(line)
  lines.add(shapeLines)
  lines.add(targetLines)
  result = lines.join("\n") & "\n"

# ------------------------------------------------------------------------------------------------------------------
# HipModel: what GpuModel + call / apply / fit do for a CompileGpu target (model.nim:21-24, 302-454)
# ------------------------------------------------------------------------------------------------------------------

type
  HipModel*[T] = ref object
    ctx*: GpuContext
    handle: ptr EgModel
    program*: Program            # the source program (model.source)

proc newHipModel*[T](source: Program, ctx: GpuContext, params, caches: Table[TensorId, Tensor[T]]): HipModel[T] =
  ## newModel for the GPU side (model.nim:215-251): compile, then upload the parameter values the host drew
  ## (parser.nim:714 initRange through newRandTensor, model.nim:241-249) so that host and device start identical.
  when T isnot float32 and T isnot float64:
    {.error: "not a valid scalar type".}   # model.nim:259
  result = HipModel[T](ctx: ctx, program: source)
  check eg_model_compile(ctx.rawHandle(), source.toKd().cstring, result.handle.addr)
  for id, tensor in params:
    check paramWrite[T](result.handle, cint(int(id)), tensor.data[0].addr, int64(tensor.len))
  for id, tensor in caches:
    check paramWrite[T](result.handle, cint(int(id)), tensor.data[0].addr, int64(tensor.len))

proc close*[T](model: HipModel[T]) =
  if not model.handle.isNil:
    check eg_model_free(model.handle)
    model.handle = nil

proc bindInputs[T](model: HipModel[T], args: openArray[(string, Tensor[T])]) =
  check eg_model_clear_inputs(model.handle)
  for (name, tensor) in args:
    if name notin model.program.inputs:
      raise RuntimeError(msg: name & " is not an input to the model")          # model.nim:358-359
    var shape = newSeq[int64](max(tensor.shape.len, 1))
    for it, size in tensor.shape:
      shape[it] = int64(size)
    check setInputHost[T](model.handle, name.cstring, tensor.data[0].addr, cint(tensor.shape.len), shape[0].addr)

proc call*[T](model: HipModel[T], target: string, args: openArray[(string, Tensor[T])] = []): Tensor[T] =
  ## Model.call (model.nim:392-406): writeInput per argument, inferShapes + allocShapes + the kernel list
  ## (eg_model_run), readOutput into a fresh tensor (gpu.nim:68-70)
  if target notin model.program.targets:
    raise RuntimeError(msg: target & " is not a target of the model")          # model.nim:395-396
  model.bindInputs(args)
  check eg_model_run(model.handle, target.cstring)
  if int(model.program.targets[target].output) != 0:
    var
      rank: cint
      shape8: array[8, int64]
    check eg_model_output_shape(model.handle, target.cstring, rank.addr, shape8[0].addr)
    var shape = newSeq[int](int(rank))
    for it in 0..<int(rank):
      shape[it] = int(shape8[it])
    result = newTensor[T](shape)
    if result.len > 0:
      check readOutput[T](model.handle, target.cstring, result.data[0].addr, int64(result.len))

proc apply*[T](model: HipModel[T], target: string, args: openArray[(string, Tensor[T])] = []) =
  ## Model.apply (model.nim:408-411) — without reading an output back
  if target notin model.program.targets:
    raise RuntimeError(msg: target & " is not a target of the model")
  model.bindInputs(args)
  check eg_model_run(model.handle, target.cstring)

proc fit*[T](model: HipModel[T], target: string, args: openArray[(string, Tensor[T])], batchSize: int = 32) =
  ## Model.fit (model.nim:413-454) as one call: epoch += 1, the data set is uploaded once, every batch is one graph launch
  if args.len == 0:
    raise RuntimeError(msg: "Model.fit requires at least one input tensor. Use Model.apply instead if the target has zero inputs.")
  var
    names = newSeq[cstring](args.len)
    data = newSeq[ptr T](args.len)
    onDevice = newSeq[cint](args.len)
    ranks = newSeq[cint](args.len)
    shapes8 = newSeq[int64](8 * args.len)
  for it, (name, tensor) in args:
    names[it] = name.cstring
    data[it] = tensor.data[0].addr
    ranks[it] = cint(tensor.shape.len)
    for dim, size in tensor.shape:
      shapes8[8 * it + dim] = int64(size)
  when T is float64:
    check eg_model_fit_f64(model.handle, target.cstring, cint(args.len), names[0].addr, data[0].addr, onDevice[0].addr,
                           ranks[0].addr, shapes8[0].addr, int64(batchSize))
  else:
    check eg_model_fit(model.handle, target.cstring, cint(args.len), names[0].addr, data[0].addr, onDevice[0].addr,
                       ranks[0].addr, shapes8[0].addr, int64(batchSize))

proc readParam*[T](model: HipModel[T], id: TensorId, into: Tensor[T]) =
  ## The device copy is the truth once a GPU target has run (the reference never copies GPU-side updates back:
  ## stateLocation only grows, model.nim:326-345)
  check paramRead[T](model.handle, cint(int(id)), into.data[0].addr, int64(into.len))

proc writeParam*[T](model: HipModel[T], id: TensorId, value: Tensor[T]) =
  check paramWrite[T](model.handle, cint(int(id)), value.data[0].addr, int64(value.len))

proc epoch*[T](model: HipModel[T]): int = int(eg_model_epoch(model.handle))
proc `epoch=`*[T](model: HipModel[T], value: int) = check eg_model_set_epoch(model.handle, int64(value))
proc keepValues*[T](model: HipModel[T], on = true) =
  ## Plans keep the values of every result tensor (the reference's own behaviour: every kernel's output is a tensor,
  ## model.nim:295-300) instead of, where all readers allow it, one predicate bit per element — for debugging
  check eg_model_keep_values(model.handle, cint(ord(on)))
proc emitIr*[T](model: HipModel[T]): string = $eg_model_plan_text(model.handle)                    # model.nim:262-264
proc launches*[T](model: HipModel[T], target: string): string = $eg_model_launch_text(model.handle, target.cstring)

proc storeState*[T](model: HipModel[T]): seq[byte] =
  ## `params` and `caches` as io/serialize.nim:348-349 writes them, read from the device copies
  var n: csize_t
  check eg_model_state_bytes(model.handle, n.addr)
  result = newSeq[byte](int(n))
  if n > 0:
    check eg_model_store_state(model.handle, result[0].addr, n, n.addr)

proc loadState*[T](model: HipModel[T], bytes: openArray[byte]): int =
  ## Returns the number of bytes consumed (what follows belongs to the caller's stream)
  var used: csize_t
  if bytes.len > 0:
    check eg_model_load_state(model.handle, bytes[0].unsafeAddr, csize_t(bytes.len), used.addr)
  result = int(used)
