# exprgrad/runtimes/hip.nim — the proc set of gpu.nim:24-52 over the C ABI of libexprgrad_hip.so (mirrors cl.nim 1:1).
# Level 1 of the integration (INTEGRATION.md); selected by -d:hip through the patch of gpu.nim:20-23 in nim/PATCHES.md.
# Written against exprgrad v1; never compiled (no Nim toolchain in this repository's build image) — every eg_* symbol it
# binds is exported by the library and exercised through the same C ABI by tests/cabi_harness.c and the Python tests.
{.passL: "-lexprgrad_hip".}
type
  GpuError* = ref object of CatchableError
  GpuDevice* = object
    index: cint
  EgCtx {.importc: "eg_ctx", header: "exprgrad_hip.h", incompleteStruct.} = object
  EgBuf {.importc: "eg_buf", header: "exprgrad_hip.h", incompleteStruct.} = object
  EgKernel {.importc: "eg_kernel", header: "exprgrad_hip.h", incompleteStruct.} = object
  GpuContext* = ref object
    handle: ptr EgCtx
  GpuBuffer* = object
    ctx: GpuContext
    size: int
    handle: ptr EgBuf
  GpuKernelSource* = object
    name*: string
    source*: string
  GpuKernel* = ref object
    ctx: GpuContext
    handle: ptr EgKernel

{.push importc, cdecl, header: "exprgrad_hip.h".}
proc eg_last_error(): cstring
proc eg_device_count(count: ptr cint): cint
proc eg_device_info(device: cint, name: cstring, nameCap: csize_t, vendor: cstring, vendorCap: csize_t,
                    version: cstring, versionCap: csize_t, isGpu: ptr cint): cint
proc eg_ctx_create(device: cint, res: ptr ptr EgCtx): cint
proc eg_buf_alloc(ctx: ptr EgCtx, bytes: csize_t, res: ptr ptr EgBuf): cint
proc eg_buf_free(buf: ptr EgBuf): cint
proc eg_buf_write(buf: ptr EgBuf, host: pointer, bytes: csize_t): cint
proc eg_buf_read(buf: ptr EgBuf, host: pointer, bytes: csize_t): cint
proc eg_buf_fill(buf: ptr EgBuf, pattern: pointer, patternBytes: csize_t): cint
proc eg_kernel_compile(ctx: ptr EgCtx, name, source: cstring, res: ptr ptr EgKernel): cint
proc eg_kernel_set_arg_buf(kernel: ptr EgKernel, index: cint, buf: ptr EgBuf): cint
proc eg_kernel_set_arg_i64(kernel: ptr EgKernel, index: cint, value: int64): cint
proc eg_kernel_set_arg_f32(kernel: ptr EgKernel, index: cint, value: float32): cint
proc eg_kernel_set_arg_f64(kernel: ptr EgKernel, index: cint, value: float64): cint
proc eg_kernel_launch(kernel: ptr EgKernel, dims: cint, groups, local: ptr int64): cint
proc eg_switches_reload(): cint
{.pop.}

proc check(status: cint) =                       # cl.nim:41-43
  if status != 0:
    raise GpuError(msg: $eg_last_error())

proc reloadSwitches*() =
  ## The library reads its environment switches (EG_NO_GRAPH, ...: DESIGN.md section 4) once, at first use; a program that
  ## changes one with putEnv afterwards calls this.  No reference counterpart (its switches are compile-time defines).
  check eg_switches_reload()

proc listDevices*(): seq[GpuDevice] =            # cl.nim:64-66
  var n: cint
  check eg_device_count(n.addr)
  for it in 0..<int(n):
    result.add(GpuDevice(index: cint(it)))

proc queryInfo(device: GpuDevice): (string, string, string, bool) =
  var
    name = newString(256)
    vendor = newString(256)
    version = newString(256)
    gpu: cint
  check eg_device_info(device.index, name.cstring, 256, vendor.cstring, 256, version.cstring, 256, gpu.addr)
  result = ($name.cstring, $vendor.cstring, $version.cstring, gpu != 0)

proc name*(device: GpuDevice): string = device.queryInfo()[0]
proc vendor*(device: GpuDevice): string = device.queryInfo()[1]
proc version*(device: GpuDevice): string = device.queryInfo()[2]
proc isGpu*(device: GpuDevice): bool = device.queryInfo()[3]

proc newGpuContext*(device: GpuDevice): GpuContext =      # cl.nim:83-93
  result = GpuContext()
  check eg_ctx_create(device.index, result.handle.addr)

proc rawHandle*(ctx: GpuContext): pointer = ctx.handle    # for hipmodel.nim (eg_model_compile takes the eg_ctx*)

proc newGpuContext*(): GpuContext =                       # cl.nim:95-99
  let devices = listDevices()
  if devices.len == 0:
    raise GpuError(msg: "Unable to find device")
  result = newGpuContext(devices[0])

proc allocBuffer*(ctx: GpuContext, size: int): GpuBuffer =        # cl.nim:101-106
  result.ctx = ctx
  result.size = size
  check eg_buf_alloc(ctx.handle, csize_t(size), result.handle.addr)

proc dealloc*(buffer: GpuBuffer) = check eg_buf_free(buffer.handle)

proc write*(buffer: GpuBuffer, data: pointer, size: int) =        # cl.nim:111-116 (size check in the library)
  check eg_buf_write(buffer.handle, data, csize_t(size))

proc write*[T](buffer: GpuBuffer, data: openArray[T]) =
  if data.len > 0:
    buffer.write(data[0].unsafeAddr, sizeof(T) * data.len)

proc fill*[T](buffer: GpuBuffer, value: T) =                      # cl.nim:122-126
  var val = value
  check eg_buf_fill(buffer.handle, val.addr, csize_t(sizeof(T)))

proc readInto*[T](buffer: GpuBuffer, data: ptr UncheckedArray[T]) =
  check eg_buf_read(buffer.handle, data[0].addr, csize_t(buffer.size))

proc readInto*[T](buffer: GpuBuffer, data: var seq[T]) =
  check eg_buf_read(buffer.handle, data[0].addr, csize_t(data.len * sizeof(T)))

proc read*[T](buffer: GpuBuffer): seq[T] =
  if buffer.size mod sizeof(T) != 0:
    raise GpuError(msg: "Buffer size is not divisible by item type size")
  if buffer.size > 0:
    result = newSeq[T](buffer.size div sizeof(T))
    check eg_buf_read(buffer.handle, result[0].addr, csize_t(buffer.size))

proc compile*(ctx: GpuContext, name, source: string): GpuKernel =  # cl.nim:149-179: build log in the error
  result = GpuKernel(ctx: ctx)
  check eg_kernel_compile(ctx.handle, name.cstring, source.cstring, result.handle.addr)

proc compile*(ctx: GpuContext, source: GpuKernelSource): GpuKernel =
  result = ctx.compile(source.name, source.source)

proc arg*[T](kernel: GpuKernel, index: int, value: T): GpuKernel =
  result = kernel
  when T is float32: check eg_kernel_set_arg_f32(kernel.handle, cint(index), value)
  elif T is float64: check eg_kernel_set_arg_f64(kernel.handle, cint(index), value)
  else: check eg_kernel_set_arg_i64(kernel.handle, cint(index), int64(value))

proc arg*(kernel: GpuKernel, index: int, buffer: GpuBuffer): GpuKernel =
  result = kernel
  check eg_kernel_set_arg_buf(kernel.handle, cint(index), buffer.handle)

proc run*(kernel: GpuKernel, groupSize, localSize: openArray[int]) =   # cl.nim:190-207
  if groupSize.len == 0:
    raise GpuError(msg: "Group size must have at least one dimension")
  if groupSize.len != localSize.len:
    raise GpuError(msg: "Dimension of group size must equal dimension of local size")
  var groups, local: array[3, int64]
  for it in 0..<groupSize.len:
    groups[it] = int64(groupSize[it]); local[it] = int64(localSize[it])
  check eg_kernel_launch(kernel.handle, cint(groupSize.len), groups[0].addr, local[0].addr)
