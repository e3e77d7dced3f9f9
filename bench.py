#!/usr/bin/env python
"""Benchmark of exprgrad's compiled-tensor hot path on MI355X (contract: DESIGN.md "Measurement").

  python bench.py --gpus 1 --steps K --warmup W
      matmul M=N=K=4096 float32 (BASELINE.json configs[1], the config the metric is quoted on):
      value = GFLOP/s; the same JSON line also carries the single-GPU numbers of the other
      configs ("extra": dense-net train step, XOR train step, conv2) measured in the same run.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
      data-parallel dense-net train step (configs[4]): batch 65536 per GPU ("weak" scaling,
      N = 8 is the 524288 global batch of the config), RCCL all-reduce of the parameter gradients
      before the gradientDescent update; value = samples/s over all ranks.

One "step" is one pass of the hot path over one batch of synthetic input already resident in HBM
when the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Roofline denominators live in ONE file shared with whoever checks the fractions:
# f32 MFMA 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz), HBM3E 8.0 TB/s.
with open(os.path.join(ROOT, "roofline_constants.json")) as _f:
    ROOFLINE = json.load(_f)
F32_MFMA_PEAK_TFLOPS = ROOFLINE["f32_mfma_peak_tflops"]
HBM_PEAK_GBS = ROOFLINE["hbm_peak_gbs"]
F64_MFMA_PEAK_TFLOPS = ROOFLINE["f64_mfma_peak_tflops"]

DENSE = dict(n_in=784, n_hidden=512, n_out=10, rate=0.01, batch=65536)
# algorithmic GEMM FLOPs per sample of the dense-net train step (SURVEY.md §8d cfg 5):
# forward 2 GEMMs, backward 3 (input gradient of the first layer is eliminated)
DENSE_FLOPS_PER_SAMPLE = 2 * (2 * 784 * 512) + 3 * (2 * 512 * 10)
XOR_BYTES_PER_SAMPLE = 72 * 4  # kernel-list traffic of the XOR train target


def source_fingerprint():
    """sha256 over the kernel and host sources of the library: what a PMC profile must have been taken on
    for its byte counts to describe the kernels this run times."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "exprgrad_amd", "csrc")
    for sub in ("kernels", "host", "."):
        d = os.path.join(base, sub)
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".hpp", ".cpp", ".h")):
                with open(os.path.join(d, name), "rb") as f:
                    h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


_TRAFFIC = None


def measured_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/summarize_profile.py: 2 x FETCH_SIZE + WRITE_SIZE); None when no profile covers the workload.
    The counters cannot be read inside this process (rocprofv3 wraps the command), so the figure is as
    old as the profile: traffic_is_stale() tells whether the sources changed since."""
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                _TRAFFIC = json.load(f)
        except (OSError, ValueError):
            _TRAFFIC = {}
    try:
        return _TRAFFIC[key]["traffic_bytes"]
    except KeyError:
        return None


def traffic_is_stale(key):
    """True when profiles/traffic.json was taken on other sources than the ones built now (or does not
    say): a kernel change without a re-profile must not pass old byte counts off as current."""
    measured_traffic(key)
    entry = _TRAFFIC.get(key)
    if not entry:
        return None
    return entry.get("source_fingerprint", _TRAFFIC.get("source_fingerprint")) != source_fingerprint()


def traffic_fields(key):
    t = measured_traffic(key)
    return {"traffic": t, "traffic_stale": traffic_is_stale(key) if t is not None else None,
            "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "matmul", "train", "xor", "conv2", "hbm"])
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch of the train/xor workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="matmul: skip the Model.call (H2D + product + D2H) figure; tools/profile.sh sets it so that "
                         "rocprofv3's per-kernel average covers the device-resident launches only (the first "
                         "Model.call launch touches fresh allocations and takes ~24 ms)")
    ap.add_argument("--native-dp", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--torch-dp", action="store_true",
                    help="N > 1: let torch.distributed all-reduce the gradient bucket instead of the C ABI's own RCCL "
                         "group (eg_dp_*, eg_model_step_dp — what a Nim host binds, the default)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1 train workload: weak = 65536 samples per GPU (N = 8 is the config's 524288 global "
                         "batch); strong = the config's global batch of 524288 divided over the N GPUs (SURVEY.md §8d)")
    return ap.parse_args()


DEPENDENT_LAUNCH_US = 1.5   # boundary between two dependent launches inside a captured graph (MI355X_MICROARCH.md)
# untimed seconds of the same step in front of the W warmup steps (Timer.run): a sustained-clock figure.  tools/profile.sh shortens it
# for its counter passes (EG_BENCH_SPINUP_S): under --pmc every dispatch is serialised and 0.2 s of a 12 us step is 16 000 of them
SPINUP_S = float(os.environ.get("EG_BENCH_SPINUP_S", "0.2"))

EVENTS_NOTE = ("kernel_ms_avg = one HIP event pair around the K timed steps / K, on the stream the kernels run on; "
               "kernel_ms_min = shortest step of a second, untimed pass of K steps with an event pair each")


def clock_fields(telemetry, achieved_tflops, clock=None):
    """The effective shader clock of the timed steps next to the fraction: counted on the device (DeviceClock), so
    clock x 65 536 FLOP/clk (256 CUs x 4 SIMDs x 64) is the matrix rate those steps could have reached and must not be
    below what they achieved.  The hwmon reading that stood here until round 4 (`sclk_mhz_timed_mean`) is a sampled
    sensor value and under-reads (2017 MHz next to 145.7 TFLOP/s = 2224 MHz worth of MFMAs): it stays in `telemetry`,
    labelled as a sensor, and no fraction is derived from it."""
    out = {}
    if clock and clock.get("mhz"):
        peak_at_clock = clock["mhz"] * 1e6 * 65536 / 1e12
        out["effective_clock_mhz"] = clock["mhz"]
        out["mfma_rate_at_effective_clock_tflops"] = round(peak_at_clock, 2)
        out["frac_of_rate_at_effective_clock"] = round(achieved_tflops / peak_at_clock, 4)
        out["effective_clock_source"] = clock["source"] + "; " + clock.get("pass", "")
    elif clock:
        out["effective_clock_error"] = clock.get("error")
    t = (telemetry or {}).get("sclk_mhz_timed")
    if t and t.get("mean"):
        out["hwmon_sclk_mhz_sensor_mean"] = t["mean"]
    return out


class Telemetry:
    """Shader clock, memory clock, socket power and temperatures of THIS GPU while a workload runs, read from the
    amdgpu hwmon files of its PCI device (no tool is spawned, nothing touches the device): the same binary ran the dense
    step 8 % apart on two boxes with the matmul kernel within 0.2 % — the line now says what the box
    was doing.  A daemon thread samples every 10 ms from the first spin-up step to the end of the timed region."""

    FILES = (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_input", 1e-6),
             ("temp_junction_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3))

    def __init__(self, torch, device_index):
        self.dir = None
        try:
            import glob
            props = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
            found = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
            self.dir = found[0] if found else None
            self.bdf = bdf
        except Exception:  # noqa: BLE001 - telemetry is optional
            self.dir = None
        self.samples, self._stop, self._thread, self._timed_from = [], None, None, None

    def mark_timed(self):
        """The samples from here on belong to the K timed steps (Timer.run calls this behind the barrier)."""
        self._timed_from = len(self.samples)

    def _read(self):
        row = {}
        for key, name, scale in self.FILES:
            try:
                with open(os.path.join(self.dir, name)) as f:
                    row[key] = int(f.read().strip()) * scale
            except (OSError, ValueError):
                pass
        return row

    def start(self):
        if not self.dir:
            return
        import threading
        self.samples, self._stop = [self._read()], threading.Event()

        def loop():
            while not self._stop.wait(0.01):
                self.samples.append(self._read())
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if not self.dir or self._thread is None:
            return {"available": False}
        self._stop.set()
        self._thread.join()
        self._thread = None
        self.samples.append(self._read())
        out = {"available": True, "pci": self.bdf, "samples": len(self.samples), "period_ms": 10,
               "covers": "spin-up + warmup + timed steps of this workload"}
        for key, _, _ in self.FILES:
            vals = [r[key] for r in self.samples if key in r]
            if vals:
                out[key] = {"min": round(min(vals), 1), "mean": round(sum(vals) / len(vals), 1), "max": round(max(vals), 1)}
        # the K timed steps alone (the last sample is taken behind the final synchronize: the device is already idle)
        if self._timed_from is not None:
            timed = self.samples[self._timed_from:-1]
            for key in ("sclk_mhz", "power_w"):
                vals = [r[key] for r in timed if key in r]
                if vals:
                    out[key + "_timed"] = {"samples": len(vals), "min": round(min(vals), 1),
                                           "mean": round(sum(vals) / len(vals), 1), "max": round(max(vals), 1)}
        return out


class DeviceClock:
    """Effective shader clock of the K timed steps, counted on the device (the hwmon reading
    `freq1_input` under-reads — 2017 MHz x 65 536 FLOP/clk = 132 TFLOP/s next to 145.7 achieved in the same steps).
    ONE wave of a probe kernel sits on a stream of its own while the steps run: it waits for a flag a marker kernel on
    the MAIN stream raises in front of the first timed step, reads s_memtime (shader cycles, MI355X_MICROARCH.md
    "s_memtime tick = shader cycle") and s_memrealtime (constant 100 MHz), sleeps until the marker behind the last
    timed step, reads both again: mean clock = d(cycles) / d(realtime) x 100 MHz, both counters read by the same wave
    on the same compute unit.  The wave sleeps (s_sleep) between polls, touches no LDS and holds 16 registers; the two
    markers are one-thread kernels outside the event pair.  Every spin is bounded by the realtime counter (8 s)."""

    PROBE = r"""
extern "C" __global__ void eg_clock_probe(long long* out, int* flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long born = __builtin_amdgcn_s_memrealtime();
  const long long limit = 800000000LL;   // 8 s at 100 MHz
  out[4] = 0;
  while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_amdgcn_s_memrealtime() - born > limit) { out[4] = 1; return; }
  }
  const long long c0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_amdgcn_s_memrealtime() - born > limit) { out[4] = 2; return; }
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  out[0] = c0; out[1] = r0; out[2] = c1; out[3] = r1;
}
"""
    MARK = r"""
extern "C" __global__ void eg_clock_mark(int* flags, long which) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(&flags[which], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
"""

    def __init__(self, main_ctx, device_index):
        import numpy as np
        import exprgrad_amd as eg
        self.np = np
        self.main = main_ctx
        self.side = eg.newGpuContext(device_index)           # a stream of its own for the probe wave
        self.flags = main_ctx.allocBuffer(8)
        self.out = self.side.allocBuffer(5 * 8)
        self.probe = self.side.compile("eg_clock_probe", self.PROBE).arg(0, self.out).arg(1, self.flags)
        self.mark = main_ctx.compile("eg_clock_mark", self.MARK).arg(0, self.flags)
        self.armed = False

    def arm(self):
        """Main stream idle (the caller has synchronised): clear the flags, start the probe wave."""
        self.flags.fill(0, self.np.int32)
        self.main.sync()
        self.probe.run([1], [64])
        self.armed = True

    def begin(self):
        if self.armed:
            self.mark.arg(1, 0).run([1], [64])

    def end(self):
        if self.armed:
            self.mark.arg(1, 1).run([1], [64])

    def result(self):
        """After the main stream has been synchronised."""
        if not self.armed:
            return None
        self.armed = False
        v = self.out.read(self.np.int64)
        if v[4] != 0 or v[3] <= v[1]:
            return {"error": f"probe wave gave up in phase {int(v[4])}"}
        cycles, ticks = int(v[2] - v[0]), int(v[3] - v[1])
        return {"mhz": round(cycles / ticks * 100.0, 1), "cycles": cycles, "realtime_ticks_100mhz": ticks,
                "source": "s_memtime / s_memrealtime read by one resident wave in front of and behind the timed steps"}


class Timer:
    """K steps bracketed by barrier + synchronize on both sides; max over ranks; per-step HIP
    events on the stream the kernels are launched on (torch's current stream == the context's)."""

    def __init__(self, torch, dist, world, stream, telemetry=None, clock=None):
        self.torch, self.dist, self.world, self.stream = torch, dist, world, stream
        self.telemetry, self.last_telemetry = telemetry, None
        self.clock, self.last_clock = clock, None

    def sync(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run_cold(self, step, steps, warmup):
        """The literal contract, nothing in front of it: W warmup steps, then K timed steps between two
        barrier + synchronize brackets, from whatever clock state the device is in after start-up (idle).  Reported as
        `value_cold` next to the sustained-clock `value` (run() below puts ~0.2 s of the same step in front)."""
        for _ in range(warmup):
            step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync()
        return time.perf_counter() - t0

    def run(self, step, steps, warmup):
        torch = self.torch
        # Untimed preparation, like the kernel builds that happen on the first call: keep the device
        # busy for ~0.2 s so the timed region runs at the sustained clock (tools/gemm_series.py: the
        # first ~20 ms after idle run up to 13 % slower while the clocks ramp; W = 5 steps of a
        # 1 ms kernel end inside that ramp).  Then the W warmup steps of the contract.
        # Host-side housekeeping first, so no idle gap separates warmup and timed region.
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        gc.collect()
        gc.disable()  # a generation-2 collection inside a 25 us step would dominate it
        # The number of spin-up steps must be the same on every rank (a step may contain a
        # collective): time 8 steps, agree on the slowest rank's estimate, derive the count from it.
        if self.telemetry:
            self.telemetry.start()
        t_probe = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        per_step = (time.perf_counter() - t_probe) / 8
        if self.world > 1:
            t = torch.tensor([per_step], device="cuda", dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            per_step = float(t.item())
        spin = min(4096, max(8, int(SPINUP_S / max(per_step, 1e-6))))
        self.last_spin = 8 + spin   # untimed steps before the W warmup steps (8 to estimate the step, then ~0.2 s of them)
        for i in range(spin):
            step()
            if i % 64 == 63:
                torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        self.sync()
        # The timed region carries ONE pair of events around all K steps: an event is a barrier packet with a
        # timestamp, and a pair between every two steps holds the queue for 10 - 17 us (rocprofv3 timeline of the
        # train step: the only gap of the step; the XOR step is 25 us long).  Per-step durations (their minimum) come
        # from a second, untimed pass of K steps with an event pair each.
        # Device-side clock of the timed steps (DeviceClock): only when the timed region is long against the two
        # one-thread marker kernels it adds (a few us each); shorter regions (the XOR step) get it from an extra pass below.
        clock_inline = self.clock is not None and per_step * steps >= 2e-3
        if clock_inline:
            self.clock.arm()
        if self.telemetry:
            self.telemetry.mark_timed()
        t0 = time.perf_counter()
        marks = []
        if clock_inline:
            self.clock.begin()
        starts[0].record(self.stream)
        for i in range(steps):
            step()
            marks.append(time.perf_counter())
        ends[0].record(self.stream)
        if clock_inline:
            self.clock.end()
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.telemetry:
            self.last_telemetry = self.telemetry.stop()
        self.last_clock = None
        if self.clock is not None:
            try:
                if not clock_inline:     # the same K steps once more, untimed, between the two markers
                    self.clock.arm()
                    self.clock.begin()
                    for i in range(steps):
                        step()
                    self.clock.end()
                    self.sync()
                self.last_clock = self.clock.result()
                if self.last_clock is not None:
                    self.last_clock["pass"] = "the timed steps" if clock_inline else "a repeat of the timed steps (region too short for markers)"
            except Exception as exc:  # noqa: BLE001 - a secondary figure
                self.last_clock = {"error": repr(exc)}
        bracket_ms = starts[0].elapsed_time(ends[0])
        for i in range(steps):
            starts[i].record(self.stream)
            step()
            ends[i].record(self.stream)
        self.sync()
        gc.enable()
        if os.environ.get("EG_BENCH_DEBUG"):
            host = [round((b - a) * 1e6) for a, b in zip([t0] + marks[:-1], marks)]
            print(f"[bench] host us/step: {host} tail sync {round((t0 + elapsed - marks[-1]) * 1e6)}", file=sys.stderr)
        if self.world > 1:
            t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ev = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
        return elapsed, bracket_ms / steps, ev[0]


# ------------------------------------------------------------------------------ CPU baselines

def cpu_baseline_matmul(n, budget_s=12.0):
    """The oracle's matmul loop nest (reference order y, it, x; the y loop split over all host
    cores exactly as builtinRunThreads splits it, model.nim:110-132) on a bounded row slice."""
    import numpy as np
    from oracle import refcpu
    refcpu.build()
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(2)
    b = rng.random((n, n), dtype=np.float32)
    rows = min(n, max(cores, 64))
    a = rng.random((rows, n), dtype=np.float32)
    threads = refcpu.thread_count(rows, n * (n + 1), cores)
    t0 = time.perf_counter()
    refcpu.sgemm(a, b, threads=threads)
    dt = time.perf_counter() - t0
    rows2 = int(min(n, max(rows, rows * budget_s / max(dt, 1e-6))))
    rows2 = max(cores, rows2 // cores * cores)
    a = rng.random((rows2, n), dtype=np.float32)
    threads = refcpu.thread_count(rows2, n * (n + 1), cores)
    t0 = time.perf_counter()
    refcpu.sgemm(a, b, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(2.0 * rows2 * n * n / dt / 1e9, 2), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"rows 0..{rows2} of the {n}^3 product, oracle/refcpu.c ref_sgemm (reference loop order, "
                      f"y split over {threads} threads), {dt:.2f} s wall"}


def cpu_baseline_train(text, batch_cpu=2048):
    """The oracle's train step (reference kernel list; contractions threaded over their independent
    outer loop like the Threads target) at a reduced batch, scaled per sample."""
    import numpy as np
    from oracle import kd
    cores = os.cpu_count() or 1
    m = kd.Model(text, threads=cores)
    rng = np.random.default_rng(5)
    for tid in m.params:
        m.params[tid][...] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    x = rng.random((batch_cpu, DENSE["n_in"]), dtype=np.float32)
    y = np.eye(DENSE["n_out"], dtype=np.float32)[rng.integers(0, DENSE["n_out"], size=batch_cpu)]
    m.apply("train", {"x": x, "y": y})
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 8.0 or reps < 1:
        m.apply("train", {"x": x, "y": y})
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(batch_cpu / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{reps} train steps at batch {batch_cpu} (oracle/kd.py + refcpu.c; contractions on {cores} "
                      f"threads, elementwise kernels single-threaded as in the reference), {dt * 1e3:.1f} ms/step"}


def cpu_baseline_conv2(budget_s=6.0):
    """configs[3] on the host: the reference's loop order n, y, f, dy, x, dx, c (conv2_naive,
    benchmarks/conv2/conv2.nim:49-55).  The reference parallelises only the image loop n (SURVEY.md
    Appendix A.4), so with N = 1 its policy is ONE core; a y-parallel all-core figure is given next to it,
    labelled as not the reference's policy.  Sample: a band of output rows (each row is the complete
    3 x 3 x 64 reduction), scaled by rows."""
    import numpy as np
    from oracle import refcpu
    refcpu.build()
    cores = os.cpu_count() or 1
    H = W = 256
    C = F = 64
    rng = np.random.default_rng(4)
    flt = (rng.random((F, 3, 3, C), dtype=np.float32) * 4 - 2).astype(np.float32)
    flops_row = 2.0 * (W - 2) * F * 9 * C

    def timed(rows, threads_y):
        img = rng.random((1, rows + 2, W, C), dtype=np.float32)
        t0 = time.perf_counter()
        refcpu.conv2_nhwc(img, flt, threads_y=threads_y)
        return time.perf_counter() - t0
    dt = timed(2, 1)
    rows = int(max(2, min(H - 2, 2 * budget_s / max(dt, 1e-6))))
    dt = timed(rows, 1)
    one = {"value": round(rows * flops_row / dt / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
           "sample": f"{rows} of the 254 output rows of configs[3], oracle/refcpu.c ref_conv2_nhwc (reference loop order), "
                     f"one thread = the reference's policy for N = 1 (only the image loop is parallel), {dt:.2f} s wall"}
    rows_all = H - 2
    dt_all = timed(rows_all, cores)
    one["all_cores_not_reference_policy"] = {
        "value": round(rows_all * flops_row / dt_all / 1e9, 2), "unit": "GFLOP/s", "cores": min(cores, rows_all),
        "sample": f"all 254 rows, the y loop split over {min(cores, rows_all)} threads (NOT what the reference does), {dt_all:.2f} s wall"}
    return one


def cpu_baseline_xor(batch=65536, budget_s=5.0):
    """configs[2] on the host: the oracle's XOR train step at the full batch.  No kernel of this step
    reaches the reference's 2^24 work-per-thread threshold (passes.nim:2415-2437), so its policy is one
    thread throughout."""
    import numpy as np
    from exprgrad_amd import examples
    from oracle import kd
    m = kd.Model(__import__("exprgrad_amd").dsl.to_program(*examples.xor_from_scratch()).to_text(), threads=1)
    rng = np.random.default_rng(3)
    for tid in m.params:
        m.params[tid][...] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    x = rng.integers(0, 2, size=(batch, 2)).astype(np.float32)
    y = (x[:, :1] != x[:, 1:]).astype(np.float32)
    m.apply("train", {"x": x, "y": y})
    for tid in m.params:   # rate 0.1 on a 65536-sample sum: keep the parameters finite between repetitions
        m.params[tid][...] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s or reps < 1:
        for tid in m.params:
            m.params[tid][...] = 0.05
        m.apply("train", {"x": x, "y": y})
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(1.0 / dt, 2), "unit": "steps/s", "samples_per_s": round(batch / dt, 1), "cores": 1, "kind": "port",
            "sample": f"{reps} train steps at the full batch of {batch} (oracle/kd.py + refinterp.c, one thread = the "
                      f"reference's policy: no kernel of the step reaches 2^24 work per thread), {dt * 1e3:.1f} ms/step"}


# ------------------------------------------------------------------------------ workloads

def _matmul_end_to_end(args, env, a, b, c, ctx, n, flops):
    """What the reference's benchmark times (matmul_gpu.nim:35-46): model.call with host tensors."""
    end_to_end = None
    if env["rank"] == 0 and n <= 8192 and not args.no_end_to_end:
        from exprgrad_amd import examples as refcases
        from exprgrad_amd import model as egm
        import numpy as np
        model = egm.compile(*refcases.matmul_graph(), gpu=ctx)
        ha, hb = a.cpu().numpy(), b.cpu().numpy()
        model.call("c", {"a": ha, "b": hb})
        t0 = time.perf_counter()
        for _ in range(3):
            hc = model.call("c", {"a": ha, "b": hb})
        dt = (time.perf_counter() - t0) / 3
        # the copies alone, for the PCIe rate they reach (same pageable arrays, same staged path)
        buf = ctx.allocTensor((n, n))
        buf.write(ha)
        t1 = time.perf_counter()
        for _ in range(3):
            buf.write(ha)
        h2d = (time.perf_counter() - t1) / 3
        out = np.empty((n, n), dtype=np.float32)
        buf.readInto(out)
        t1 = time.perf_counter()
        for _ in range(3):
            buf.readInto(out)
        d2h = (time.perf_counter() - t1) / 3
        buf.buffer.dealloc()
        mib = n * n * 4 / 2 ** 20
        end_to_end = {"ms_per_call": round(dt * 1e3, 2), "gflops": round(flops / dt / 1e9, 1),
                      "bytes_moved_mib": round(3 * mib, 1), "pcie_gbs_over_the_call": round(3 * n * n * 4 / dt / 1e9, 1),
                      "h2d_gbs": round(n * n * 4 / h2d / 1e9, 1), "d2h_gbs": round(n * n * 4 / d2h / 1e9, 1),
                      "note": "Model.call with pageable host arrays (what benchmarks/matmul/matmul_gpu.nim:35-46 times): H2D of A "
                              "and B, the product, D2H of C; copies staged through pinned buffers by a thread pool (csrc/host_copy.cpp)",
                      "checksum_matches_device_result": bool(np.allclose(hc[:8, :8], c[:8, :8].cpu().numpy(), rtol=1e-5))}
        model.close()
    return end_to_end


def run_matmul(args, env):
    torch, ops, ctx, timer = env["torch"], env["ops"], env["ctx"], env["timer"]
    n = args.size
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2 + env["rank"])
    a = torch.rand((n, n), device="cuda", dtype=torch.float32, generator=gen)  # U[0,1): matmul_gpu.nim:69-70
    b = torch.rand((n, n), device="cuda", dtype=torch.float32, generator=gen)
    c = torch.empty((n, n), device="cuda", dtype=torch.float32)
    flops = 2.0 * n * n * n
    ops.sgemm(ctx, n, n, n, a, n, b, n, c, n)          # first call: module load, workspace
    timer.sync()
    time.sleep(0.25)                                    # idle: the clocks fall back, as they are when a process starts
    cold_s = timer.run_cold(lambda: ops.sgemm(ctx, n, n, n, a, n, b, n, c, n), args.steps, args.warmup)
    elapsed, ev_avg, ev_min = timer.run(lambda: ops.sgemm(ctx, n, n, n, a, n, b, n, c, n), args.steps, args.warmup)
    # ONE clock for `value` and the roofline fraction: the wall time of the K timed steps (barrier + synchronize on both
    # sides).  HIP events are reported next to it (kernel_ms_avg / kernel_ms_min, frac_by_events; EVENTS_NOTE).
    achieved = flops * args.steps / elapsed / 1e12
    achieved_events = flops / (ev_avg * 1e-3) / 1e12
    # what the reference's benchmark times (matmul_gpu.nim:35-46): model.call with host tensors — 128 MiB
    # host->device, the product, 64 MiB device->host per call.  Reported next to the kernel figure, never as `value`.
    end_to_end = None
    try:
        end_to_end = _matmul_end_to_end(args, env, a, b, c, ctx, n, flops)
    except Exception as exc:  # noqa: BLE001 - a secondary figure
        end_to_end = {"error": repr(exc)}
    return {
        "end_to_end": end_to_end,
        "metric": "GFLOP/s matmul 4096^3 f32 (1 GPU)" if n == 4096 else f"GFLOP/s matmul {n}^3 f32",
        "value": round(flops * args.steps * env["world"] / elapsed / 1e9, 1), "unit": "GFLOP/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "value_cold": {"value": round(flops * args.steps * env["world"] / cold_s / 1e9, 1), "unit": "GFLOP/s",
                       "ms_per_step": round(cold_s / args.steps * 1e3, 4),
                       "frac": round(flops * args.steps / cold_s / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                       "note": "the literal --warmup W / --steps K figure from an idle device (no spin-up in front): the "
                               "first ~20 ms after idle run inside the clock ramp"},
        "config": {"workload": f"matmul M=N=K={n} float32 (BASELINE configs[1]): C = A*B through eg_sgemm "
                               "(MFMA + LDS tiled HIP kernel), A, B ~ U[0,1) resident in HBM",
                   "parallelism": "single" if env["world"] == 1 else f"{env['world']} independent replicas",
                   "timed_steps": args.steps, "sustained_clock_spinup_s": SPINUP_S,
                   "spinup_note": "value is a sustained-clock figure: ~0.2 s of the same launch run untimed before the W "
                                  "warmup steps (spinup_steps), because W = 5 steps of a 1 ms kernel end inside the clock ramp"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), **clock_fields(timer.last_telemetry, achieved, timer.last_clock),
                     **(traffic_fields("matmul4096") if n == 4096 else {"traffic": None}),
                     "kernel": "eg::gemm::gemm_f32_mfma_kernel<256,256,32,128,64,NN,DMA> (skewed waves, 32-deep k-tiles)", "flops_per_launch": flops,
                     "clock": "wall time of the timed steps (the clock `value` uses)",
                     "kernel_ms_avg": round(ev_avg, 4), "kernel_ms_min": round(ev_min, 4), "events": EVENTS_NOTE,
                     "frac_by_events": round(achieved_events / F32_MFMA_PEAK_TFLOPS, 4)},
        "spinup_steps": timer.last_spin, "telemetry": timer.last_telemetry,
    }


def run_matmul_sizes(args, env):
    """Square products below and around the headline size through the same entry point (eg_sgemm picks tile, k-tile depth
    and slicing per shape): what a user's own layer sizes get.  Same clock discipline as every other figure of the line
    (Timer.run: ~0.2 s of the same launch, then W warmup steps, then the timed steps)."""
    torch, ops, ctx, timer = env["torch"], env["ops"], env["ctx"], env["timer"]
    steps = max(args.steps, 50)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    rows = {}
    for n in (512, 1024, 1280, 1536, 1792, 2048, 3072):
        a = torch.rand((n, n), device="cuda", dtype=torch.float32, generator=gen)
        b = torch.rand((n, n), device="cuda", dtype=torch.float32, generator=gen)
        c = torch.empty((n, n), device="cuda", dtype=torch.float32)
        elapsed, ev_avg, _ = timer.run(lambda: ops.sgemm(ctx, n, n, n, a, n, b, n, c, n), steps, args.warmup)
        tflops = 2.0 * n * n * n * steps / elapsed / 1e12
        rows[str(n)] = {"us_per_launch": round(elapsed / steps * 1e6, 2), "tflops": round(tflops, 2),
                        "frac_of_mfma_peak": round(tflops / F32_MFMA_PEAK_TFLOPS, 4), "kernel_us_by_events": round(ev_avg * 1e3, 2)}
    return {"metric": "TFLOP/s of C = A*B, M = N = K, float32, through eg_sgemm", "unit": "TFLOP/s", "sizes": rows,
            "timed_steps": steps, "sustained_clock_spinup_s": SPINUP_S,
            "roofline": {"bound": "mfma", "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "achieved": rows["1024"]["tflops"], "frac": rows["1024"]["frac_of_mfma_peak"],
                         "kernel": "1024^3: gemm_pair_kernel<64,64,32,32,NN,64-deep k-tiles> (one block per CU, two waves per "
                                   "sub-tile); 512^3 likewise; 1536^3 - 3072^3: gemm_f32_mfma_kernel<64,64,...>, four waves, "
                                   "up to four blocks per CU",
                         "traffic": None}}


def run_hbm_kernels(args, env):
    """The bandwidth-bound library kernels of SURVEY.md 8(a)-K at cfg-5 sizes (65 536 x 512 floats = 134 MB per tensor; the
    skinny 65 536 x 10 forms next to them): elementwise maps and their derived gradients (dnn.nim:26-40), bias add
    (dnn.nim:19-24), column / row / full reductions (bias gradient, softmax sums, loss: base.nim:57-67), axpy
    (gradientDescent, base.nim:37-38).  Since round 2 every model fuses these away, so they are timed here through the
    library entry points themselves: algorithmic bytes (each operand once) / mean launch time by HIP events on the launch
    stream, against the 8 TB/s HBM peak (the guide's float4-copy ceiling is 6.29 TB/s)."""
    torch, ops, ctx = env["torch"], env["ops"], env["ctx"]
    stream = env["timer"].stream
    rows_n, cols_n = 65536, 512
    n = rows_n * cols_n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    # FOUR operand sets in rotation (1.6 GB): a 134 MB operand would otherwise be served by the 256 MB Infinity Cache on
    # every launch after the first (MI355X_MICROARCH.md "Infinity Cache") and the figure would not be an HBM rate
    SETS = 4
    xs_, gs_, ys_ = [], [], []
    for _ in range(SETS):
        xs_.append(torch.rand((rows_n, cols_n), device="cuda", generator=gen) - 0.5)
        gs_.append(torch.rand((rows_n, cols_n), device="cuda", generator=gen) - 0.5)
        ys_.append(torch.empty((rows_n, cols_n), device="cuda"))
    bias = torch.rand((cols_n,), device="cuda", generator=gen)
    colv, rowv, tot = torch.empty((cols_n,), device="cuda"), torch.empty((rows_n,), device="cuda"), torch.empty((1,), device="cuda")
    xs = torch.rand((rows_n, 10), device="cuda", generator=gen) - 0.5
    ys = torch.empty_like(xs)
    bias_s, col_s = torch.rand((10,), device="cuda", generator=gen), torch.empty((10,), device="cuda")
    steps = max(args.steps, 48)
    turn = [0]

    def timed(step):
        def go():
            i = turn[0] = (turn[0] + 1) % SETS
            step(xs_[i], gs_[i], ys_[i])
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.05 or k < 8:   # sustained clocks, like every other figure of the line
            go()
            k += 1
            if k % 32 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            go()
        b.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps * 1e3   # us per launch

    f = 4
    cases = [
        ("map_relu", lambda x, g, y: ops.map_(ctx, "relu", n, x, y), 2 * n * f),
        ("map_sigmoid", lambda x, g, y: ops.map_(ctx, "sigmoid", n, x, y), 2 * n * f),
        ("map_tanh", lambda x, g, y: ops.map_(ctx, "tanh", n, x, y), 2 * n * f),
        ("map_grad_relu", lambda x, g, y: ops.map_grad(ctx, "relu", n, x, g, y), 3 * n * f),
        ("map_grad_sigmoid", lambda x, g, y: ops.map_grad(ctx, "sigmoid", n, x, g, y), 3 * n * f),
        ("map_grad_tanh", lambda x, g, y: ops.map_grad(ctx, "tanh", n, x, g, y), 3 * n * f),
        ("bias_add", lambda x, g, y: ops.bias_add(ctx, rows_n, cols_n, bias, y, accumulate=True), 2 * n * f),
        ("colsum", lambda x, g, y: ops.colsum(ctx, rows_n, cols_n, x, colv), n * f),
        ("rowsum", lambda x, g, y: ops.rowsum(ctx, rows_n, cols_n, x, rowv), n * f + rows_n * f),
        ("sum", lambda x, g, y: ops.total(ctx, n, x, tot), n * f),
        ("axpy", lambda x, g, y: ops.axpy(ctx, n, -0.01, g, y), 3 * n * f),
        ("fill", lambda x, g, y: ops.fill(ctx, n, 0.0, y), n * f),
        ("bias_add_65536x10", lambda x, g, y: ops.bias_add(ctx, rows_n, 10, bias_s, ys, accumulate=True), 2 * rows_n * 10 * f),
        ("colsum_65536x10", lambda x, g, y: ops.colsum(ctx, rows_n, 10, xs, col_s), rows_n * 10 * f),
        ("rowsum_65536x10", lambda x, g, y: ops.rowsum(ctx, rows_n, 10, xs, rowv), rows_n * 11 * f),
    ]
    out = {}
    for name, step, nbytes in cases:
        for y in ys_:
            y.zero_()
        us = timed(step)
        gbs = nbytes / (us * 1e-6) / 1e9
        out[name] = {"us": round(us, 2), "mb": round(nbytes / 1e6, 2), "gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    big = [k for k in out if not k.endswith("x10")]
    worst = min(big, key=lambda k: out[k]["frac"])
    return {"metric": "GB/s of the bandwidth-bound library kernels (eg_map / eg_map_grad / eg_bias_add / eg_colsum / eg_rowsum / "
                      "eg_sum / eg_axpy / eg_fill_f32), 65536 x 512 float32 operands (134 MB each) and the 65536 x 10 forms",
            "unit": "GB/s", "kernels": out, "timed_steps": steps, "sustained_clock_spinup_s": 0.05,
            "bytes": "algorithmic: every operand read once, every result written once (an accumulating result also read); four "
                     "operand sets (1.6 GB) in rotation, so that no launch finds its operands in the 256 MB Infinity Cache",
            "clock": "one HIP event pair around the timed launches on the launch stream / launches",
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": out[worst]["gbs"],
                         "frac": out[worst]["frac"], "kernel": worst + " (the slowest of the 134 MB forms)",
                         "copy_ceiling_gbs": 6290, **traffic_fields("hbm_kernels")}}


def run_float64(args, env):
    """compile[float64] (model.nim:253-260), the scalar type of several of the reference's own tests and of its conv2
    benchmark: eg_dgemm at three sizes against the float64 matrix peak (v_mfma_f64_16x16x4_f64: 32 FLOP / clk / SIMD,
    measured ceiling 77.8 TFLOP/s, tools/mfma_ceiling_f64.hip), and the reference's conv2 benchmark program in its own type
    and shape (benchmarks/conv2/conv2.nim:128-138, 330-364: image 960 x 1280 x 8, 8 filters of 3 x 3 x 8, float64) through
    eg_model_run on device-resident inputs — the direct float64 convolution (kernels/conv2_direct.cpp: 16 pixels x 4 taps x
    16 filter columns per matrix instruction, input rows staged in LDS), 157 MB per call against 1.4 GFLOP."""
    import ctypes
    import numpy as np
    from exprgrad_amd import _lib, examples
    from exprgrad_amd import model as egm
    torch, ctx, timer = env["torch"], env["ctx"], env["timer"]
    steps = max(args.steps, 20)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    rows = {}
    for n in (1024, 2048, 4096):
        a = torch.rand((n, n), device="cuda", dtype=torch.float64, generator=gen) - 0.5
        b = torch.rand((n, n), device="cuda", dtype=torch.float64, generator=gen) - 0.5
        c = torch.empty((n, n), device="cuda", dtype=torch.float64)

        def step():
            _lib.call("eg_dgemm", ctx.handle, 0, 0, n, n, n, ctypes.c_void_p(a.data_ptr()), n, ctypes.c_void_p(b.data_ptr()), n,
                      ctypes.c_void_p(c.data_ptr()), n, 0, None)
        # The clock probe's resident wave (DeviceClock: 16 registers of one SIMD) takes the slot of a 128-register wave with it:
        # eg_dgemm's blocks fill the register files exactly (four waves per SIMD), the probe's CU then holds one block instead
        # of two and 1024 tiles need a third round — 4096^3 runs 2.48 instead of 2.25 ms next to the probe
        # (tools/f64_only.py).  So the rows are timed WITHOUT the probe; a second pass with it only reads the clock.
        probe, timer.clock = timer.clock, None
        try:
            elapsed, ev_avg, _ = timer.run(step, steps, args.warmup)
        finally:
            timer.clock = probe
        tflops = 2.0 * n * n * n * steps / elapsed / 1e12
        rows[str(n)] = {"us_per_launch": round(elapsed / steps * 1e6, 2), "tflops": round(tflops, 2),
                        "frac_of_f64_mfma_peak": round(tflops / F64_MFMA_PEAK_TFLOPS, 4), "kernel_us_by_events": round(ev_avg * 1e3, 2)}
        if probe is not None and n == 4096:
            timer.run(step, steps, args.warmup)
            clock = timer.last_clock
            if clock and clock.get("mhz"):   # the float64 matrix rate at that clock: 32 FLOP / clk / SIMD x 1024 SIMDs
                at_clock = clock["mhz"] * 1e6 * 32768 / 1e12
                rows[str(n)].update({"effective_clock_mhz_separate_pass": clock["mhz"],
                                     "frac_of_rate_at_effective_clock": round(tflops / at_clock, 4)})
    out = {"metric": "TFLOP/s of C = A*B, M = N = K, float64, through eg_dgemm", "unit": "TFLOP/s", "sizes": rows, "timed_steps": steps,
           "roofline": {"bound": "mfma", "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": rows["4096"]["tflops"],
                        "frac": rows["4096"]["frac_of_f64_mfma_peak"], "traffic": None,
                        "kernel": "dgemm_kernel<128,128,2,4,...> (eight waves of 64 x 32 per tile, two blocks per CU)"}}
    H, W, C, F = 960, 1280, 8, 8
    model = egm.compile(*examples.conv2_3d(), gpu=ctx, dtype=np.float64)
    image = torch.rand((H, W, C), device="cuda", dtype=torch.float64, generator=gen)
    filters = torch.rand((F, 3, 3, C), device="cuda", dtype=torch.float64, generator=gen) * 4 - 2
    feed = {"image": image, "filters": filters}
    elapsed, ev_avg, _ = timer.run(lambda: model.apply("conv2", feed), steps, args.warmup)
    flops = 2.0 * (H - 2) * (W - 2) * F * 9 * C
    nbytes = 8.0 * (H * W * C + (H - 2) * (W - 2) * F + F * 9 * C)
    out["conv2_benchmark"] = {
        "workload": "benchmarks/conv2/conv2.nim:330-364: 960 x 1280 x 8 image, 8 filters 3 x 3 x 8, float64, compile[float64] + call",
        "ms_per_call": round(elapsed / steps * 1e3, 4), "gflops": round(flops * steps / elapsed / 1e9, 1),
        "algorithmic_gbs": round(nbytes * steps / elapsed / 1e9, 1), "frac_of_hbm_peak": round(nbytes * steps / elapsed / 1e9 / HBM_PEAK_GBS, 4),
        # 18 matrix instructions per 16 pixels whatever F <= 16 is: the instruction-bound time of this shape is 36 us
        "matrix_instruction_bound_ms": round((H - 2) * (W - 2) / 16.0 * 18 * 64 / (1024 * 2.4e9) * 1e3, 4),
        "kernel": "eg_conv_mfma64_c8_f8_3x3 (v_mfma_f64_16x16x4_f64, a ring of 4 input row segments in LDS per 128-pixel column strip)",
        "launches": model.launch_plan("conv2").strip().splitlines()}
    model.close()
    return out


def cpu_baseline_float64(budget_s=4.0):
    """The oracle's compile[float64] form of the reference's conv2 benchmark program on a crop of its image (one core)."""
    import numpy as np
    from exprgrad_amd import dsl, examples
    from oracle import kd
    prog = dsl.to_program(*examples.conv2_3d())
    prog.scalar = "f64"
    m = kd.Model(prog.to_text())
    rng = np.random.default_rng(0)
    H, W = 96, 1280
    image, filters = rng.random((H, W, 8)), rng.random((8, 3, 3, 8)) * 4 - 2
    t0, calls = time.perf_counter(), 0
    while calls < 1 or time.perf_counter() - t0 < budget_s:
        m.call("conv2", {"image": image, "filters": filters})
        calls += 1
    dt = (time.perf_counter() - t0) / calls
    flops = 2.0 * (H - 2) * (W - 2) * 8 * 72
    return {"value": round(flops / dt / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"conv2 benchmark program, float64, a 96 x 1280 x 8 crop of the 960 x 1280 x 8 image, {calls} call(s) of the oracle's interpreter"}


def build_dense(env, batch):
    from exprgrad_amd import examples as refcases
    from exprgrad_amd import model as egm
    from exprgrad_amd.parallel import DataParallel, GpuEngine
    torch = env["torch"]
    model = egm.compile(*refcases.dense_softmax_net(DENSE["n_in"], DENSE["n_hidden"], DENSE["n_out"], DENSE["rate"]),
                        gpu=env["ctx"])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)  # identical parameters on every rank
    for tid in model.params.ids():
        shape = model._param_shapes[tid]
        model.params[tid] = (torch.rand(shape, device="cuda", generator=gen) * 0.2 - 0.1).cpu().numpy()
    gen.manual_seed(100 + env["rank"])  # a different shard of synthetic data per rank
    x = torch.rand((batch, DENSE["n_in"]), device="cuda", dtype=torch.float32, generator=gen)
    labels = torch.randint(0, DENSE["n_out"], (batch,), device="cuda", generator=gen)
    y = torch.nn.functional.one_hot(labels, DENSE["n_out"]).to(torch.float32).contiguous()
    env["dp_path"] = "none"
    dp = None
    if env["world"] > 1 and env.get("native_dp"):
        # the library's own communicator (what a Nim host binds): rank 0 draws the id, torch.distributed
        # only carries its 128 bytes.  Every rank must take the same path, so a failure anywhere
        # (librccl missing, init error) sends all of them to the torch.distributed exchange instead.
        import torch.distributed as dist
        from exprgrad_amd.parallel import NativeDataParallel, RcclGroup
        ok, group = 1, None
        # ncclCommInitRank blocks until every rank has joined: a rank that cannot even load RCCL must be found
        # BEFORE anyone enters it.  Probe on every rank (dlopen + ncclGetUniqueId, no communication), agree, then init.
        try:
            own_id = RcclGroup.unique_id()
        except Exception as exc:  # noqa: BLE001
            ok, own_id = 0, None
            env["dp_fallback_reason"] = repr(exc)
        probe = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(probe, op=dist.ReduceOp.MIN)
        if int(probe.item()) == 1:
            try:
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if env["rank"] == 0:
                    uid.copy_(torch.frombuffer(bytearray(own_id), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                group = RcclGroup(env["ctx"], bytes(uid.cpu().numpy().tobytes()), env["rank"], env["world"])
            except Exception as exc:  # noqa: BLE001
                ok = 0
                env["dp_fallback_reason"] = repr(exc)
        else:
            ok = 0
            env.setdefault("dp_fallback_reason", "another rank could not load RCCL")
        flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            dp = NativeDataParallel(model, "train", group, reduction="mean")
            dp.engine = _NativeEngine(model, "train")
            env["dp_path"] = "native"
            env["rccl_ranks"] = group.rccl_count()      # what RCCL says (ncclCommCount), not the argument echoed back
            env["rccl_group"] = group
        elif group is not None:
            group.close()
    if dp is None:
        dp = DataParallel(GpuEngine(model, "train"), reduction="mean")
        if env["world"] > 1:
            import torch.distributed as dist
            env["dp_path"] = "torch"
            # ranks an RCCL communicator spans: torch's nccl backend is RCCL; under gloo (one-GPU test mode) no RCCL ran
            env["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
    return model, dp, x, y


class _NativeEngine:
    """The single-GPU reference step of run_train for the --native-dp path (library-owned bucket)."""

    def __init__(self, model, target):
        self.model, self.target = model, target

    def set_grad_scale(self, s):
        self.model.set_grad_scale(s)

    def run_backward(self, args):
        self.model.run_backward(self.target, args)

    def run_update(self):
        self.model.run_update(self.target)


def run_train(args, env):
    world = env["world"]
    strong = getattr(args, "scaling", "weak") == "strong"
    batch = args.batch or (DENSE["batch"] * 8 // world if strong else DENSE["batch"])
    model, dp, x, y = build_dense(env, batch)
    inputs = [("x", x), ("y", y)]
    single = None
    if world > 1:
        # the same step on this GPU's shard without the gradient exchange, in the same run: the
        # 1-GPU point of the scaling series (the N = 1 invocation reports matmul as its primary metric)
        eng = dp.engine

        def alone():
            eng.set_grad_scale(1.0)
            eng.run_backward(inputs)
            eng.run_update()
        k1 = min(args.steps, 20)
        t1, _, _ = env["timer"].run(alone, k1, 3)
        single = {"value": round(batch * k1 / t1, 1), "unit": "samples/s", "ms_per_step": round(t1 / k1 * 1e3, 4),
                  "note": "one GPU's shard, no all-reduce, slowest rank; N-GPU value / (N x this) = scaling efficiency"}
    if world == 1:
        # one GPU: the step is Model.apply (one captured launch sequence); the split form backward | exchange |
        # update only exists for the exchange and costs a second graph boundary (~15 us)
        model._bind_all(inputs)
        step = lambda: model.apply("train", inputs)
    else:
        step = lambda: dp.step(inputs)
    elapsed, ev_avg, ev_min = env["timer"].run(step, args.steps, args.warmup)
    telemetry, device_clock = env["timer"].last_telemetry, env["timer"].last_clock
    exchange = None
    if world > 1:
        # One run decides whether exchanging the early gradients under the last long contraction pays: the same step
        # with the split allowed and forbidden, and the bare all-reduce of the bucket, each over min(steps, 20) steps
        # AFTER the timed region (value / ms_per_step above are the default configuration's).
        import torch.distributed as dist
        torch = env["torch"]
        k2 = min(args.steps, 20)
        exchange = {}
        group = env.get("rccl_group")
        if group is not None:
            for name, allowed in (("step_ms_unsplit", False), ("step_ms_split", True)):
                group.set_split(allowed)
                t2, _, _ = env["timer"].run(step, k2, 3)
                exchange[name] = round(t2 / k2 * 1e3, 4)
                exchange["pieces_" + name[8:]] = group.last_pieces()
            bucket = torch.zeros(max(model.grad_bucket("train")[1], 1), device="cuda")
            t3, _, _ = env["timer"].run(lambda: group.all_reduce(bucket), k2, 3)
        else:
            bucket = dp.engine.bucket
            t3, _, _ = env["timer"].run(lambda: dist.all_reduce(bucket, op=dist.ReduceOp.SUM), k2, 3)
        exchange["allreduce_us"] = round(t3 / k2 * 1e6, 2)
        exchange["allreduce_floats"] = int(bucket.numel())
    samples = batch * world * args.steps
    step_flops = DENSE_FLOPS_PER_SAMPLE * batch
    achieved = step_flops * args.steps / elapsed / 1e12           # the clock `value` uses (see run_matmul)
    achieved_events = step_flops / (ev_avg * 1e-3) / 1e12
    out = {
        "metric": "train samples/s dense net 784-512-10 (data parallel)",
        "value": round(samples / elapsed, 1), "unit": "samples/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "steps_per_s": round(args.steps / elapsed, 2),
        "config": {"workload": "dense(784,512)-relu-dense(512,10)-softmax-crossEntropy-gradientDescent(0.01) train "
                               "step (BASELINE configs[4]), synthetic x ~ U[0,1), one-hot labels, forward + "
                               "backward + parameter-gradient all-reduce + update",
                   "global_batch": batch * world, "per_gpu_batch": batch,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "collective": {"none": "none",
                                  "native": "RCCL all-reduce from the C ABI (eg_model_step_dp, the library's own communicator)",
                                  "torch": ("gloo all-reduce (EG_BENCH_ONE_GPU test mode)" if os.environ.get("EG_BENCH_ONE_GPU") == "1"
                                            else "RCCL all-reduce through torch.distributed (nccl backend)")}[env.get("dp_path", "none")],
                   "rccl_ranks": env.get("rccl_ranks", 0),
                   "scaling": "strong" if strong else "weak",
                   "grad_bucket_floats": model.grad_bucket("train")[1],
                   "timed_steps": args.steps, "sustained_clock_spinup_s": SPINUP_S},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), **clock_fields(telemetry, achieved, device_clock),
                     **(traffic_fields("train") if batch == DENSE["batch"] else {"traffic": None}),
                     "kernel": "whole train step on one GPU (5 contractions dominate: gemm_f32_mfma_kernel)",
                     "flops_per_launch": step_flops, "clock": "wall time of the timed steps (the clock `value` uses)",
                     "kernel_ms_avg": round(ev_avg, 4), "kernel_ms_min": round(ev_min, 4), "events": EVENTS_NOTE,
                     "frac_by_events": round(achieved_events / F32_MFMA_PEAK_TFLOPS, 4)},
        "spinup_steps": env["timer"].last_spin, "telemetry": telemetry,
    }
    if single:
        out["single_gpu_reference"] = single
        # the N-GPU value over N times the one-GPU value of the same step measured in this run (weak scaling: per-GPU
        # work is the same in both); 1 -> N speed-up = N x this.  With --scaling strong the one-GPU point runs the
        # shard's batch too, so the figure is the exchange's cost only and the line says which.
        out["scaling_efficiency"] = round(out["value"] / (world * single["value"]), 4)
        out["speedup_vs_single_gpu_reference"] = round(out["value"] / single["value"], 3)
    if exchange:
        out["exchange"] = exchange
    if env.get("dp_fallback_reason"):
        out["config"]["native_dp_fallback"] = env["dp_fallback_reason"]
    out["scaling"] = "strong" if strong else "weak"
    return out, model


def run_xor(args, env):
    from exprgrad_amd import examples as refcases
    from exprgrad_amd import model as egm
    torch = env["torch"]
    batch = args.batch or 65536
    model = egm.compile(*refcases.xor_from_scratch(), gpu=env["ctx"])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    x = torch.randint(0, 2, (batch, 2), device="cuda", generator=gen).to(torch.float32)
    y = (x[:, :1] != x[:, 1:]).to(torch.float32).contiguous()
    inputs = [("x", x), ("y", y)]
    steps = max(args.steps, 50)
    elapsed, ev_avg, ev_min = env["timer"].run(lambda: model.apply("train", inputs), steps, args.warmup)
    gbs = XOR_BYTES_PER_SAMPLE * batch * steps / elapsed / 1e9      # the clock `value` uses (see run_matmul)
    return {"telemetry": env["timer"].last_telemetry, "metric": "train steps/s XOR net (examples/xor_from_scratch) batch 65536", "value": round(steps / elapsed, 1),
            "unit": "steps/s", "samples_per_s": round(batch * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
            "config": {"workload": f"examples/xor_from_scratch train step, batch {batch} (BASELINE configs[2]), Model.apply on "
                                   f"device-resident inputs; timed steps = max(--steps, 50) = {steps}",
                       "timed_steps": steps, "sustained_clock_spinup_s": SPINUP_S},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4),
                         **(traffic_fields("xor") if batch == 65536 else {"traffic": None}),
                         "kernel": "whole step = ONE launch (eg_rows: 15 forward / backward kernels per sample in registers; the last block to "
                                   "arrive folds the partial rows and runs the four update kernels); launch- and hand-off-latency bound",
                         "bytes_per_launch": XOR_BYTES_PER_SAMPLE * batch, "clock": "wall time of the timed steps",
                         "kernel_ms_avg": round(ev_avg, 4)},
            "scaling_note": "launch-bound: the step is one launch of ~8 us around 1 MB of real traffic, so its "
                            "time is launch and hand-off latency, not bandwidth.  Under data parallelism a step adds one 17-float "
                            "all-reduce (tens of us) and cannot get shorter: steps/s does NOT scale with GPUs.  The only "
                            "claim this workload supports is weak scaling in samples/s (65536 samples per GPU: N GPUs "
                            "process N x 65536 samples in one step time + one all-reduce latency); the north-star's "
                            ">= 6x from 1 to 8 GPUs is a statement about the dense step (configs[4]), whose 1.1 ms of "
                            "matrix work dwarfs the exchange."}


def run_xor_dp(args, env):
    """BASELINE configs[2] data parallel (weak: 65536 samples per GPU): the 17-float gradient bucket makes the
    step a pure latency test of the all-reduce — reported for completeness next to the dense step."""
    from exprgrad_amd import examples as refcases
    from exprgrad_amd import model as egm
    from exprgrad_amd.parallel import DataParallel, GpuEngine
    torch = env["torch"]
    batch = 65536
    model = egm.compile(*refcases.xor_from_scratch(), gpu=env["ctx"])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    for tid in model.params.ids():
        model.params[tid] = (torch.rand(model._param_shapes[tid], device="cuda", generator=gen) * 0.2 - 0.1).cpu().numpy()
    gen.manual_seed(300 + env["rank"])
    x = torch.randint(0, 2, (batch, 2), device="cuda", generator=gen).to(torch.float32)
    y = (x[:, :1] != x[:, 1:]).to(torch.float32).contiguous()
    dp = DataParallel(GpuEngine(model, "train"), reduction="sum")   # the XOR loss is a plain sum (xor_from_scratch.nim:28)
    inputs = [("x", x), ("y", y)]
    steps = max(args.steps, 50)
    elapsed, ev_avg, _ = env["timer"].run(lambda: dp.step(inputs), steps, args.warmup)
    world = env["world"]
    return {"metric": "train steps/s XOR net (data parallel, 65536 samples per GPU)", "value": round(steps / elapsed, 1),
            "unit": "steps/s", "samples_per_s": round(batch * world * steps / elapsed, 1),
            "ms_per_step": round(elapsed / steps * 1e3, 4), "grad_bucket_floats": model.grad_bucket("train")[1]}


def run_conv2(args, env):
    torch, ops, ctx = env["torch"], env["ops"], env["ctx"]
    N, H, W, C, F, FH, FW = 1, 256, 256, 64, 64, 3, 3  # BASELINE configs[3]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    img = torch.rand((N, H, W, C), device="cuda", generator=gen)                 # conv2.nim:337
    flt = torch.rand((F, FH, FW, C), device="cuda", generator=gen) * 4 - 2       # conv2.nim:338
    out = torch.empty((N, H - FH + 1, W - FW + 1, F), device="cuda")
    steps = max(args.steps, 50)
    elapsed, ev_avg, ev_min = env["timer"].run(
        lambda: ops.conv2_nhwc(ctx, N, H, W, C, F, FH, FW, img, flt, out), steps, args.warmup)
    flops = 2.0 * N * (H - FH + 1) * (W - FW + 1) * F * FH * FW * C
    telemetry, device_clock = env["timer"].last_telemetry, env["timer"].last_clock
    achieved = flops * steps / elapsed / 1e12                          # the clock `value` uses (see run_matmul)
    # the two gradients derive makes of conv2 (same FLOP count each), for the record
    gout = torch.rand(out.shape, device="cuda", generator=gen) - 0.5
    gflt, gimg = torch.empty_like(flt), torch.empty_like(img)
    _, gf_avg, _ = env["timer"].run(
        lambda: ops.conv2_nhwc_grad_filter(ctx, N, H, W, C, F, FH, FW, img, gout, gflt), steps, args.warmup)
    _, gi_avg, _ = env["timer"].run(
        lambda: ops.conv2_nhwc_grad_image(ctx, N, H, W, C, F, FH, FW, flt, gout, gimg), steps, args.warmup)
    backward = {"grad_filter_ms": round(gf_avg, 4), "grad_filter_tflops": round(flops / (gf_avg * 1e-3) / 1e12, 2),
                "grad_image_ms": round(gi_avg, 4), "grad_image_tflops": round(flops / (gi_avg * 1e-3) / 1e12, 2)}
    return {"telemetry": telemetry, "metric": "GFLOP/s conv2 3x3 256x256x64->64 f32", "value": round(flops * steps / elapsed / 1e9, 1),
            "unit": "GFLOP/s", "ms_per_step": round(elapsed / steps * 1e3, 4),
            "config": {"workload": f"benchmarks/conv2 3x3, 1x256x256x64 -> 64 filters float32 NHWC (BASELINE configs[3]) through "
                                   f"eg_conv2_nhwc; timed steps = max(--steps, 50) = {steps}",
                       "timed_steps": steps, "sustained_clock_spinup_s": SPINUP_S},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), **clock_fields(telemetry, achieved, device_clock),
                         **traffic_fields("conv2"),
                         "kernel": "conv2_halo_kernel<9,3,3> (LDS-resident 10x34 halo, 8x32 patch x 64 filters per block)", "flops_per_launch": flops,
                         "clock": "wall time of the timed steps", "kernel_ms_avg": round(ev_avg, 4),
                         "frac_by_events": round(flops / (ev_avg * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)},
            "backward": backward}


def run_fashion_fit(args, env):
    """One epoch of Model.fit on the reference's flagship network (examples/fashion_mnist/fashion_mnist.nim:
    39-57: reshape-conv2-leakyRelu-maxpool2 x2, dense, softmax, crossEntropy, adam) with synthetic data of
    the data set's shape, at the reference's default batch size of 32 and at 4096."""
    import numpy as np
    from exprgrad_amd import examples as refcases
    from exprgrad_amd import model as egm
    torch = env["torch"]
    model = egm.compile(*refcases.fashion_mnist_net(), gpu=env["ctx"])
    rng = np.random.default_rng(6)
    samples = 60000
    x = rng.random((samples, 784), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, samples)]
    out = {"metric": "Model.fit samples/s, fashion_mnist network (reference flagship example), one epoch of 60000",
           "unit": "samples/s", "data": "synthetic 28x28 images in host arrays; the upload of the data set (once per epoch, "
                   "overlapped with the batches) is inside the timed region"}
    for batch in (32, 4096):
        model.fit("fit", {"x": x[:batch * 4], "y": y[:batch * 4]}, batch_size=batch)   # builds, captures
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit("fit", {"x": x, "y": y}, batch_size=batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"batch_{batch}"] = {"value": round(samples // batch * batch / dt, 1), "epoch_ms": round(dt * 1e3, 2),
                                 "us_per_batch": round(dt / (samples // batch) * 1e6, 1)}
        # what bounds a batch: the launches of the captured sequence are dependent (each reads what its predecessor
        # wrote), so a batch cannot be shorter than launches x the dependent-launch boundary of the device
        # (MI355X_MICROARCH.md: ~1.5 us inside a graph), nor than its bytes / HBM rate
        plan = [ln for ln in model.launch_plan("fit").splitlines() if ln.startswith("[")]
        tensor_bytes = 4.0 * batch * (784 + 24 * 24 * 8 * 2 + 12 * 12 * 8 * 3 + 10 * 10 * 16 * 2 + 5 * 5 * 16 * 3 + 10 * 4)
        out[f"batch_{batch}"]["bound"] = {
            "library_calls_per_batch": len(plan) + 1, "dependent_launch_floor_us": round((len(plan) + 1) * DEPENDENT_LAUNCH_US, 1),
            "algorithmic_mb_per_batch": round(2 * tensor_bytes / 1e6, 2),
            "hbm_floor_us": round(2 * tensor_bytes / (HBM_PEAK_GBS * 1e9) * 1e6, 2),
            "kernels_per_batch": len(plan) + 1,
            "sample_group": any("sample-fused" in ln for ln in plan),
            "measured_us_per_dependent_kernel": 4.5,
            "measured_note": "the forward + backward pass of a step is ONE kernel with one block per sample (DESIGN.md section 3 'Round 5'); "
                             "round 6: its five convolution members run on the matrix cores (16 x 16 x 4, gathered fragments) and the "
                             "parameters are staged in LDS once (36.7 -> 28.3 us per batch-32 step), then, under cycle stamps (EG_SAMPLE_TRACE), batched "
                             "staging loads, literal trip counts, four-wide gathers, no barrier between independent members: 21.4 us on one box "
                             "(tools/fit_sample_ab.sh); the "
                             "launch chain it replaces, EG_NO_SAMPLE_FUSE=1: 16 kernels of 4.1 - 8.0 us each, 71 us.  A dependent kernel "
                             "ends >= 4.5 us after its predecessor here whatever it does; the floor above uses the guide's 1.5 us per boundary.",
            "note": "calls = launches of the plan + the segment copy of the batch's rows; some calls are two kernels (a "
                    "k-sliced contraction and its fixed-order sum).  Bytes: every activation / gradient of the network "
                    "written once and read once, float32.  At batch 32 the step is launch-bound, at 4096 "
                    "it is the sum of its kernels (profiles/README.md lists them)."}
    out["value"] = out["batch_32"]["value"]
    out["roofline"] = {"bound": "launch", "achieved": out["batch_32"]["us_per_batch"], "unit": "us per batch-32 step",
                       "peak": out["batch_32"]["bound"]["dependent_launch_floor_us"],
                       "frac": round(out["batch_32"]["bound"]["dependent_launch_floor_us"] / out["batch_32"]["us_per_batch"], 3),
                       "traffic": None,
                       "note": "floor / achieved: the batch-32 step against its dependent-launch floor.  The step is three launches and "
                               "its time is the sample kernel's (~15 of 21 us): 26 members, 22 block barriers, each member a pass over "
                               "LDS-resident tensors; the fraction says how far the step is from a chain of three empty kernels"}
    model.close()
    return out


def cpu_baseline_fit(budget_s=5.0):
    """Model.fit of the fashion_mnist network on the host: the oracle's kernel list run batch by batch the way
    model.nim:413-454 does (Model.epoch bumped ONCE per fit call, model.nim:436 — not per batch; results zeroed per batch,
    model.nim:447-449), batch 32; a pass over the 8 batches of the sample is one fit call.  No kernel of a 32-sample batch
    reaches the reference's 2^24 work-per-thread threshold: one thread is its policy."""
    import numpy as np
    from exprgrad_amd import dsl, examples
    from oracle import kd
    m = kd.Model(dsl.to_program(*examples.fashion_mnist_net()).to_text(), threads=1)
    rng = np.random.default_rng(6)
    for tid in m.params:
        m.params[tid][...] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    batch = 32
    x = rng.random((batch * 8, 784), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, batch * 8)]
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s or n < 2:
        i = (n % 8) * batch
        if i == 0:
            m.epoch += 1      # a new pass over the data = a new fit call (what fit.cpp does on the device)
        m.apply("fit", {"x": x[i:i + batch], "y": y[i:i + batch]})
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(batch / dt, 1), "unit": "samples/s", "cores": 1, "kind": "port", "interpreted": True,
            "sample": f"{n} batches of 32 (oracle/kd.py: contractions on refcpu.c's compiled loop nest, every other kernel — the "
                      f"convolutions and their gradients among them — on refinterp.c's kernel INTERPRETER, one thread = the "
                      f"reference's policy at this batch size), {dt * 1e3:.2f} ms per batch.  The reference compiles these loops "
                      f"with LLVM; an interpreter is several times slower, so this figure is a lower bound of the reference's rate"}


def compile_latency():
    """compile[float32] on this backend (model.nim:215-251, 270-273): eg_model_compile + the first run of the train
    target (plan-time kernels: fusion groups, generated-epilogue contractions), for the XOR, dense and fashion_mnist
    models — cold (empty code-object cache) and warm (the cache the cold run left), each in a process of its own."""
    import subprocess
    import tempfile
    tool = os.path.join(ROOT, "tools", "compile_time.py")
    out = {}
    with tempfile.TemporaryDirectory() as cache:
        # comgr keeps a code-object cache of its own (~/.cache/comgr) which the models benchmarked above have
        # already filled in this process's box: point it at the empty directory too, or "cold" is not cold
        env = dict(os.environ, EG_KERNEL_CACHE=cache, AMD_COMGR_CACHE_DIR=os.path.join(cache, "comgr"))
        for name in ("cold", "warm"):
            done = subprocess.run([sys.executable, tool], env=env, capture_output=True, text=True, timeout=600)
            if done.returncode != 0:
                return {"error": done.stderr[-500:]}
            out[name] = json.loads(done.stdout.strip().splitlines()[-1])
    compiler = out["cold"].pop("compiler")
    out["warm"].pop("compiler", None)
    return {"compiler": compiler, **out,
            "note": "seconds per model: eg_model_compile, then the first and second run of the train target at the config's batch"}


def per_config_block(line, extra):
    """Compact per-config figures INSIDE `roofline` (the driver keeps that dict whole but only a 2 000-character tail of
    the line, so extra.* does not survive it): one entry per BASELINE config and per widening, each with its time, the
    fraction of its bound and which bound that is."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    pc = {"cfg2_matmul4096": {"ms": line.get("ms_per_step"), "frac": get(line, "roofline", "frac"), "bound": "mfma",
                              "effective_clock_mhz": get(line, "roofline", "effective_clock_mhz")}}
    t = extra.get("train", {})
    if "error" not in t and t:
        pc["cfg5_step"] = {"ms": t.get("ms_per_step"), "frac": get(t, "roofline", "frac"), "bound": "mfma",
                           "effective_clock_mhz": get(t, "roofline", "effective_clock_mhz"),
                           "frac_at_clock": get(t, "roofline", "frac_of_rate_at_effective_clock")}
    x = extra.get("xor", {})
    if "error" not in x and x:
        pc["cfg3_xor"] = {"us": round(x["ms_per_step"] * 1e3, 2), "frac": get(x, "roofline", "frac"), "bound": "hbm (18.87 MB/step)"}
    c = extra.get("conv2", {})
    if "error" not in c and c:
        flops = get(c, "roofline", "flops_per_launch") or 0.0
        def fr(ms):
            return round(flops / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4) if ms else None
        pc["cfg4_conv2"] = {"bound": "mfma", "fwd_us": round(c["ms_per_step"] * 1e3, 2), "fwd": get(c, "roofline", "frac"),
                            "gradf_us": round(get(c, "backward", "grad_filter_ms") * 1e3, 2), "gradf": fr(get(c, "backward", "grad_filter_ms")),
                            "gradi_us": round(get(c, "backward", "grad_image_ms") * 1e3, 2), "gradi": fr(get(c, "backward", "grad_image_ms"))}
    f = extra.get("fashion_mnist_fit", {})
    if "error" not in f and f:
        pc["fit_b32"] = {"us_per_batch": get(f, "batch_32", "us_per_batch"), "frac": get(f, "roofline", "frac"), "bound": "launch"}
        pc["fit_b4096"] = {"us_per_batch": get(f, "batch_4096", "us_per_batch"),
                           "hbm_floor_us": get(f, "batch_4096", "bound", "hbm_floor_us")}
    m = extra.get("matmul_sizes", {})
    if "error" not in m and m:
        pc["matmul_sizes_frac"] = {k: v["frac_of_mfma_peak"] for k, v in m.get("sizes", {}).items()}
    h = extra.get("hbm_kernels", {})
    if "error" not in h and h:
        pc["hbm_kernels_frac_of_8TBs"] = {k: v["frac"] for k, v in h.get("kernels", {}).items()}
    d = extra.get("float64", {})
    if "error" not in d and d:
        pc["float64"] = {"dgemm4096_frac": get(d, "roofline", "frac")}
    return pc


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per
    GPU, torch.distributed.run on 127.0.0.1 with a free port) and pass their exit code on.  Under a launcher
    (WORLD_SIZE set) the world must be the N that was asked for: the line never reports another n_gpus than --gpus."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch {args.gpus} ranks "
                  f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}), or run "
                  f"`python bench.py --gpus {args.gpus}` without a launcher and it starts them itself", file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    launch_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, (world, args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (one-GPU boxes): EG_BENCH_ONE_GPU=1 puts every rank on cuda:0 and exchanges over gloo,
    # which exercises the N > 1 code path (sharding, barrier, max over ranks) without RCCL
    one_gpu = os.environ.get("EG_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import exprgrad_amd as eg
    from exprgrad_amd import ops

    # a side stream (not the legacy NULL stream) so the backend can capture its launch sequences
    # into HIP graphs; torch allocations, events and the RCCL collective all follow it
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = eg.newGpuContext(local_rank, stream=stream.cuda_stream)
    # N > 1: the C ABI's own RCCL group unless --torch-dp; the one-GPU test mode has no second device for RCCL
    native = world > 1 and not args.torch_dp and not one_gpu
    env = {"torch": torch, "ops": ops, "ctx": ctx, "world": world, "rank": rank, "native_dp": native,
           "timer": Timer(torch, dist, world, stream, Telemetry(torch, local_rank) if rank == 0 else None)}
    if rank == 0 and world == 1 and os.environ.get("EG_BENCH_NO_DEVICE_CLOCK") != "1":
        try:
            env["timer"].clock = DeviceClock(ctx, local_rank)
        except Exception as exc:  # noqa: BLE001 - a secondary figure
            print(f"[bench] device clock probe unavailable: {exc!r}", file=sys.stderr)

    workload = args.workload
    if workload == "auto":
        workload = "matmul" if world == 1 else "train"

    model = None
    if workload == "matmul":
        line = run_matmul(args, env)
    elif workload == "train":
        line, model = run_train(args, env)
    elif workload == "xor":
        line = run_xor(args, env)
    elif workload == "hbm":
        line = run_hbm_kernels(args, env)
        line["value"] = line["roofline"]["achieved"]
    else:
        line = run_conv2(args, env)

    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": line.get("scaling", "weak"), "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    line = {**{k: line[k] for k in ("metric", "value", "unit")}, **base,
            **{k: v for k, v in line.items() if k not in ("metric", "value", "unit")}}

    if world > 1 and workload == "train" and not args.no_extra:
        # secondary figure; never allowed to take the primary line down with it (every rank runs it:
        # it contains collectives)
        try:
            small = argparse.Namespace(**{**vars(args), "steps": min(args.steps, 50), "warmup": 3, "batch": 0})
            line["extra"] = {"xor": run_xor_dp(small, env)}
        except Exception as exc:  # noqa: BLE001
            line["extra"] = {"xor": {"error": repr(exc)}}
    if rank == 0 and world == 1:
        if not args.no_extra and args.workload == "auto":
            # single-GPU numbers of the other BASELINE configs, same run (reference point for the
            # 1 -> N scaling of the data-parallel step: compare extra.train.value with the --gpus N value)
            extra = {}
            small = argparse.Namespace(**{**vars(args), "steps": min(args.steps, 20), "warmup": 3, "batch": 0})

            def guarded(name, fn):
                # a secondary figure must never take the primary line down with it
                try:
                    extra[name] = fn()
                except Exception as exc:  # noqa: BLE001
                    extra[name] = {"error": repr(exc)}

            def train():
                nonlocal model
                out, model = run_train(small, env)
                return out
            guarded("train", train)
            if "config" in extra.get("train", {}):
                extra["train"]["config"]["workload"] += f"; extra.* figures time min(--steps, 20) = {small.steps} steps"
            guarded("xor", lambda: run_xor(small, env))
            guarded("conv2", lambda: run_conv2(small, env))
            guarded("fashion_mnist_fit", lambda: run_fashion_fit(small, env))
            guarded("matmul_sizes", lambda: run_matmul_sizes(small, env))
            guarded("hbm_kernels", lambda: run_hbm_kernels(small, env))
            guarded("float64", lambda: run_float64(small, env))
            guarded("compile_latency", compile_latency)
            line["extra"] = extra
            try:
                line["roofline"]["per_config"] = per_config_block(line, extra)
            except Exception as exc:  # noqa: BLE001 - a summary of figures that are all in extra.*
                line["roofline"]["per_config"] = {"error": repr(exc)}
            if not args.no_cpu_baseline:
                # every config gets its CPU figure, bounded to a few seconds each
                def baseline(name, fn):
                    if "error" in extra.get(name, {"error": 1}):
                        return
                    try:
                        extra[name]["cpu_baseline"] = fn()
                    except Exception as exc:  # noqa: BLE001
                        extra[name]["cpu_baseline"] = {"error": repr(exc)}
                baseline("train", lambda: cpu_baseline_train(model.source_text))
                baseline("xor", cpu_baseline_xor)
                baseline("conv2", cpu_baseline_conv2)
                baseline("fashion_mnist_fit", cpu_baseline_fit)
                baseline("matmul_sizes", lambda: cpu_baseline_matmul(1024, budget_s=2.0))
                baseline("float64", cpu_baseline_float64)
            if "error" not in extra["train"]:
                # the --gpus N > 1 invocations report the data-parallel train step; its 1-GPU point:
                line["scaling_series_n1"] = {"metric": extra["train"]["metric"], "value": extra["train"]["value"],
                                             "unit": extra["train"]["unit"], "ms_per_step": extra["train"]["ms_per_step"]}
        if not args.no_cpu_baseline:
            try:
                if workload == "matmul":
                    line["cpu_baseline"] = cpu_baseline_matmul(args.size)
                elif workload == "train":
                    line["cpu_baseline"] = cpu_baseline_train(model.source_text)
            except Exception as exc:  # noqa: BLE001 - report, keep the measured line
                line["cpu_baseline"] = {"error": repr(exc)}
    if rank == 0:
        try:  # which compiler built the generated kernels of this run (csrc/rtc.cpp)
            import ctypes
            from exprgrad_amd import _lib
            buf = ctypes.create_string_buffer(512)
            _lib.call("eg_compiler_info", buf, 512)
            line["runtime_compiler"] = buf.value.decode()
        except Exception as exc:  # noqa: BLE001
            line["runtime_compiler"] = repr(exc)
        # ONE JSON line; the long secondary blocks come first so that the last 2 KB of the line (what a log tail keeps) hold
        # the headline: value, roofline with per_config, cpu_baseline
        head = ("metric", "value", "unit")
        tail_keys = ("n_gpus", "steps", "warmup", "ms_per_step", "value_cold", "higher_is_better", "scaling", "vs_baseline",
                     "dtype", "data", "config", "scaling_series_n1", "cpu_baseline", "roofline")
        line = {**{k: line[k] for k in head}, **{k: v for k, v in line.items() if k not in head and k not in tail_keys},
                **{k: line[k] for k in tail_keys if k in line}}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
