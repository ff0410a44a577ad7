#!/usr/bin/env python
"""bench.py — headline benchmark of the sink hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload ns|config3|config4|config5|ns_time]

Default workload (the pipeline the metric and north_star name, at config 3's size):
    Mix(Signal(sin, ω=1kHz), Signal(noise[26 460 000 x 8], 44.1 kHz)) |> Until(600 s)
        |> Filt(Bandstop, 0.5 kHz, 2 kHz) |> ToFramerate(48 kHz) |> sink
The reference's rewrite rules (src/filters.jl:143-148) move the resampler under the filter, so
the engine runs K3 (polyphase resampler with the Mix fused into its staging) and then K2 (order-10
SOS IIR at 48 kHz).  A step is one `so_plan_execute` of that tree with the noise leaf and the
result both resident in HBM.  BASELINE config 3 (the Filt-less resampler run)
is timed in the same process and reported as the "config3" object of the same JSON line.

N>1: one process per GPU (torch.distributed / RCCL).  Started under `torch.distributed.run` (RANK / WORLD_SIZE set) the
process is one rank; started plainly with `--gpus N` it launches N fresh children of itself -- before anything touches a
GPU -- and relays rank 0's line.  `ns` / `config3`: every rank sinks its own independent signal (weak scaling, no
collective); at N > 1 over RCCL the default line also carries a "config4" object (the sharded Append with its device-
to-device gather).  `config4`: the 64 scenes of an Append are
sharded over the ranks and the result slabs are all-gathered device to device.  `config5`: each
rank owns a 128-channel slab of the 1024-channel signal; no exchange step.

One JSON line is printed by rank 0 (driver contract) with these extra objects:
  roofline        the dominant stage of the headline workload against the HBM roofline: its
                  algorithmic bytes / its hipEvent time inside the library, on the stream it runs on
  roofline_sink   the whole sink from the TIMED LOOP: algorithmic bytes of the sink / ms_per_step
  stages          every stage of the plan (hipEvent ms, launches, algorithmic bytes)
  parity_gate     engine vs CPU oracle on a prefix of the same input: checked after the timed loops and
                  before anything is printed (a failure exits without a result line)
  parity_full     the timed plan's own result against the oracle on the same noise, whole length
  cpu_baseline    the CPU oracle (a port of the reference's block-pull engine) rebuilt here with
                  -O3 -march=native, timed on this host: 1 thread, and all channels in parallel
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_GBS = 6290.0   # measured copy ceiling quoted by the same guide
METRIC = "frames/sec sink() 44.1kHz 8ch Mix+Filt+Resample; achieved HBM GB/s"


# ----------------------------------------------------------------------------- trees
def tree_ns(so, noise, n_in):
    return (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(noise, 44.1 * so.kHz)) | so.Until(n_in * so.frames)
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))


def tree_config3(so, noise, n_in):
    v = os.environ.get("SIGOPS_BENCH_PLAIN")  # tuning aid (not a reported workload): resampler alone
    if v:
        return so.Signal(noise, 44.1 * so.kHz) | so.Until(n_in * so.frames) | so.ToFramerate(48 * so.kHz)
    return (so.Signal(noise, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz))
            | so.Until(n_in * so.frames) | so.ToFramerate(48 * so.kHz))


def scene(so, noise, k, n):
    """config 4, scene k"""
    return (so.Mix(so.Signal(so.sin, ω=(500 + 25 * k) * so.Hz) | so.Until(n * so.frames), so.Signal(noise, 44.1 * so.kHz))
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms))


def tree_config5(so, x):
    return so.Signal(x, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)


# ----------------------------------------------------------------------------- helpers
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def lib_sha16():
    import hashlib

    p = os.path.join(ROOT, "signaloperators.jl_amd", "csrc", "libsigops.so")
    if not os.path.exists(p):
        return None
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def pmc_traffic(stage_name):
    """HBM bytes per execute of a stage from the committed rocprofv3 PMC passes of this command (the newest
    profiles/rNN/bench_pmc_hbm.json, written by tools/collect_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs,
    corrected as MI355X_MICROARCH.md prescribes).  A file read, NOT a measurement of this run -- and only quoted while
    the kernel sources it was collected on (their hash is taken BEFORE the passes and kept next to the counters) are the
    ones in the tree: after any change to them the line carries null and says so, instead of a number gone stale."""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "bench_pmc_hbm.json")), reverse=True)
    if not paths:
        return None, None
    rel = os.path.relpath(paths[0], ROOT)
    with open(paths[0]) as f:
        d = json.load(f)
    if d.get("kernel_sources_sha16") != kernel_sources_sha16():
        return None, "%s is from other kernel sources (%s, now %s): not quoted" % (rel, d.get("kernel_sources_sha16"), kernel_sources_sha16())
    for key, val in d.get("stages", {}).items():
        if stage_name.startswith(key):
            return val.get("corrected_bytes_per_execute"), rel + " (rocprofv3 --pmc passes of this command on these kernel sources; not this run)"
    return None, None


def pmc_traffic_sink(tag):
    """HBM bytes per execute of a WHOLE sink (every stage of workload `tag` in the newest profiles/rNN/bench_pmc_hbm.json added up):
    for the lines whose roofline is the sink's, not one kernel's (config 4, config 5).  Same rules as pmc_traffic."""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "bench_pmc_hbm.json")), reverse=True)
    if not paths:
        return None, None
    rel = os.path.relpath(paths[0], ROOT)
    with open(paths[0]) as f:
        d = json.load(f)
    if d.get("kernel_sources_sha16") != kernel_sources_sha16():
        return None, "%s is from other kernel sources (%s, now %s): not quoted" % (rel, d.get("kernel_sources_sha16"), kernel_sources_sha16())
    vals = [v.get("corrected_bytes_per_execute") for k, v in d.get("stages", {}).items() if k.startswith(tag + ":")]
    if not vals:
        return None, None
    return float(sum(vals)), rel + " (all stages of this workload, rocprofv3 --pmc passes on these kernel sources; not this run)"


def kernel_sources_sha16():
    """hash of the device code's sources (what the PMC numbers depend on; the .so itself differs from build to build)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "signaloperators.jl_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "k*.h")) + [os.path.join(d, "sigops_internal.h")]):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def self_launch(n, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: N children of this script, one rank each (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set), started BEFORE this process imports torch or touches a device; rank 0's JSON line is
    relayed.  The children are polled: the first one that exits non-zero (an import error, a rendezvous port taken
    between the bind below and the child's own bind, a rank that failed) ends the others -- they would otherwise sit in
    a barrier or a collective until the rendezvous / RCCL watchdog gives up --, and so does an overall time limit
    (SIGOPS_BENCH_TIMEOUT seconds, default 3600).  No re-exec anywhere: fresh children, or a non-zero exit."""
    import socket
    import tempfile

    if timeout_s is None:
        timeout_s = float(os.environ.get("SIGOPS_BENCH_TIMEOUT", "3600"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()  # (a file, not a pipe: nobody has to drain it while the children are polled)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    t0 = time.monotonic()
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            why = "rank %d exited with %d" % bad[0]
        elif time.monotonic() - t0 > timeout_s:
            why = "no result after %.0f s (SIGOPS_BENCH_TIMEOUT)" % timeout_s
        if why:
            for p in procs:  # (exactly the children started above, by handle)
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
            break
        time.sleep(0.2)
    out0.seek(0)
    for line in out0.read().decode().splitlines():
        if line.startswith("{") and not why:
            print(line, flush=True)
    if why or any(codes):
        raise SystemExit("bench.py: %s; ranks exited with %s" % (why or "a rank failed", codes))


def launch_path_only(args, torch, dist, rank, world):
    """No HIP device (a build container): the engine has no CPU path, so nothing can be timed -- what runs is the N-rank
    path around it (rendezvous, barriers, max over ranks, rank 0 prints), with a line that says so."""
    t = torch.zeros(1, dtype=torch.float64)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    el = time.perf_counter() - t0
    if dist is not None:
        t[0] = el
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                          "data": "none", "valid": False,
                          "config": {"workload": "NO HIP DEVICE: launch path only (rendezvous over gloo, barriers, max over ranks); "
                                                 "the sink engine has no CPU path and nothing was measured",
                                     "ranks_seen": dist.get_world_size() if dist is not None else 1},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def timed_loop(plan, optr, stream, steps, warmup, torch, dist, dev, series=None, profile=False):
    """W untimed executes, then exactly K timed ones between barrier + synchronize pairs.  `series`
    (a list) receives the device time of every execute, warm-ups first: events recorded on the launch
    stream between the executes, read after the timed region (no synchronisation inside it)."""
    nev = min(warmup + steps, 64) + 2 if series is not None else 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nev)]
    k = 0
    if profile:
        # deferred profiling (so_plan_set_profiling(plan, 2)): the library records its per-kernel events
        # during the executes and never synchronises.  Switched on before the warm-ups (the events are
        # created by the first execute) and restarted, a counter reset, right before the timed region.
        plan.set_profiling(2)
    if nev:
        ev[0].record()
    for _ in range(warmup):
        plan.execute(optr, stream)
        if k + 1 < nev:
            k += 1
            ev[k].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if profile:
        plan.set_profiling(2)  # (restart: the means are over the timed executes only)
    gap = -1
    if k + 1 < nev:  # (restart the event chain: the host's synchronize + barrier above is device idle time)
        k += 1
        ev[k].record()
        gap = k - 1
    host = []
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.execute(optr, stream)
        if k + 1 < nev:
            k += 1
            ev[k].record()
        host.append(time.perf_counter())
    t_issued = time.perf_counter()
    t_lastev = None
    if nev and k >= 1:
        ev[k].synchronize()  # the last execute's event: the launch stream has drained
        t_lastev = time.perf_counter()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if series is not None:
        series.extend(round(ev[i].elapsed_time(ev[i + 1]), 4) for i in range(k) if i != gap)
        # host side of the same loop: when each execute call returned, and when the last one had
        series.append({"host_return_ms": [round((h - t0) * 1e3, 3) for h in host[:64]],
                       "host_issue_total_ms": round((t_issued - t0) * 1e3, 3),
                       "launch_stream_drained_ms": round((t_lastev - t0) * 1e3, 3) if t_lastev else None,
                       "wall_ms": round(elapsed * 1e3, 3)})
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def stage_means(plan):
    """mean hipEvent time of every step over the executes of the timed region (deferred profiling:
    events on the launch stream, recorded inside the timed loop, read here after it)"""
    acc = plan.steps()
    plan.set_profiling(False)
    for a in acc:
        a["GBps"] = a["algorithmic_bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0.0
    return acc


def stage_times(plan, optr, stream, reps=5):
    """hipEvent time of every step (library-internal events on the launch stream), averaged"""
    plan.set_profiling(True)
    acc = None
    for _ in range(reps):
        plan.execute(optr, stream)
        st = plan.steps()
        if acc is None:
            acc = [dict(s, ms=0.0) for s in st]
        for a, s in zip(acc, st):
            a["ms"] += s["ms"] / reps
    plan.set_profiling(False)
    for a in acc:
        a["GBps"] = a["algorithmic_bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0.0
    return acc


def roofline_of(stage, traffic=None, source=None):
    return {"bound": "hbm", "achieved": stage["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": stage["GBps"] / HBM_PEAK_GBS, "frac_of_copy_ceiling": stage["GBps"] / HBM_COPY_GBS,
            "traffic": traffic, "traffic_source": source, "kernel": stage["name"], "kernel_ms": stage["ms"],
            "launches": stage["launches"], "kernel_algorithmic_bytes": stage["algorithmic_bytes"]}


FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD's FP64 matrix figure for MI355X (256 CUs x 4 SIMDs x 32 FMA / clock x 2.4 GHz); the guide under
                               # /opt/skills has no fp64 row.
FP64_MFMA_MEASURED_TFLOPS = 68.9  # what back-to-back v_mfma_f64_16x16x4_f64 reach chip-wide, operands in registers (the clock drops to
                                  # 2.0 - 2.1 GHz under that load): tools/micro/mfma64_shapes.hip, profiles/r06/mfma64_shapes.txt (64.7 - 68.9)


def mfma_roofline_of(stage, n_out, nch, mfmas_per_block):
    """The fused resampler + IIR kernel is bound by the fp64 matrix pipe, not by HBM: the MFMA
    flops its launch issues for the outputs it stores -- `mfmas_per_block` v_mfma_f64_16x16x4 (2 048 flops each; the
    plan's own count, so_plan_counter: window k-steps + 14) per block of 16 outputs x 16 rows, warm-up blocks not
    counted -- over the kernel's measured time.  Reported NEXT TO `roofline` (which stays the HBM one the contract's
    per-unit figure is stated in)."""
    if not stage["name"].startswith("k_rsos") or not mfmas_per_block or mfmas_per_block <= 0:
        return None
    blocks = n_out * nch / 256.0
    tflops = blocks * mfmas_per_block * 2048 / (stage["ms"] * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_MFMA_PEAK_TFLOPS,
            "peak_measured": FP64_MFMA_MEASURED_TFLOPS, "frac_of_measured_peak": tflops / FP64_MFMA_MEASURED_TFLOPS,
            "kernel": stage["name"], "kernel_ms": stage["ms"], "mfmas_per_block": mfmas_per_block,
            "note": "issued fp64 MFMA flops of the stored outputs (%d MFMAs per 16 x 16 block) / kernel time; peak = AMD's FP64 matrix figure, peak_measured = a loop of nothing but these MFMAs on this chip (profiles/r06/mfma64_shapes.txt)"
                    % mfmas_per_block}


def one_shot_probe():
    """tools/oneshot_probe.py in fresh child processes (this process has every cache warm by now): once without a cache
    directory, twice with a new one (the second of those finds the accumulator replay on disk).  Never fails the bench."""
    import tempfile

    def run(env_extra):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SIGOPS_CACHE_DIR")}
        env.update(env_extra)
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oneshot_probe.py")], env=env, stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, timeout=180)
            for line in out.stdout.decode().splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    imp[0] = d.get("import_ms")
                    return d["calls"]
        except Exception:
            pass
        return None

    imp = [None]
    cold = run({})
    with tempfile.TemporaryDirectory() as d:
        run({"SIGOPS_CACHE_DIR": d})
        warm = run({"SIGOPS_CACHE_DIR": d})
    return {"what": "the headline sink once, in a fresh process: plan create + first execute + destroy, host clock (ms); "
                    "device-resident leaf and result; `again`: the same call repeated in that process; `import_ms`: the "
                    "package import in front of it (torch already imported), which opens the engine library -- not in one_shot_ms",
            "no_cache_dir": cold[0] if cold else None, "warm_cache_dir": warm[0] if warm else None,
            "again_in_process": cold[-1] if cold else None, "import_ms": imp[0]}


def parity_gate(so, tree_fn, noise_host, tol=1e-6):
    """engine vs oracle on a prefix of the same input (BASELINE.md §2's correctness gate).  Runs AFTER the
    timed loops: freeing this one-shot plan's device buffers right before them stalled the device for
    40-70 ms somewhere in the next ~100 ms on one run in three (round 3, profiles/r03/stall_probe.txt: 0 of 5
    runs without the gate or with a 1 s pause after it, 5 of 12 with it directly in front)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_bridge import oracle_sink, relerr

    n = noise_host.shape[0]
    x = tree_fn(so, noise_host, n)
    got = so.sink(x)[0]
    want = oracle_sink(x)
    err = float(relerr(got, want))
    res = {"relerr": err, "tolerance": tol, "input_frames": int(n), "output_frames": int(got.shape[0]),
           "oracle": "CPU restatement of the reference (DSP.jl phase-accumulator positions), tests/oracle_bridge.py"}
    if not (got.shape == want.shape and err <= tol):
        raise SystemExit(f"bench.py: parity gate failed (no result is reported): {res}")
    return res


def cpu_baseline(so, tree_fn, seconds, nch, ndt, noise_host=None, gpu_result=None):
    """The oracle (kind "port": the Julia reference itself cannot run here) on the same workload,
    rebuilt -O3 -march=native on THIS host.  Only this leg of bench.py (and the gate) touches oracle/.
    noise_host / gpu_result: the GPU run's own input (host copy) and result -- the oracle then filters the
    SAME noise and the whole overlap is compared (returned as the second value)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    native = os.path.join(ROOT, "oracle", "_native", "libsigops_oracle.so")
    flags = "-O3 -march=native"
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        native = None
    import ctypes

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bridge as ob

    if native and os.path.exists(native):
        ob.ORACLE_SO = native
        ob._lib = None
    else:
        flags = "-O3 -march=x86-64-v3 (native rebuild failed)"
    n_in = int(round(seconds * 44100))
    if noise_host is not None:
        n_in = min(n_in, noise_host.shape[0])
        noise = np.asfortranarray(noise_host[:n_in])
    else:
        rng = np.random.default_rng(1983)
        noise = np.asfortranarray(rng.standard_normal((n_in, nch)).astype(ndt))
    x = tree_fn(so, noise, n_in)
    t0 = time.perf_counter()
    y = ob.oracle_sink(x)
    dt1 = time.perf_counter() - t0
    full = None
    if gpu_result is not None:
        # the oracle saw the first n_in input frames: its last outputs miss inputs the GPU run had
        # (resampler look-ahead) -- unless it saw everything
        whole = n_in == noise_host.shape[0]
        m = y.shape[0] if whole else max(0, y.shape[0] - 4096)
        err = float(ob.relerr(gpu_result[:m], y[:m]))
        full = {"relerr": err, "frames_compared": int(m), "channels": int(nch), "whole_workload": bool(whole),
                "what": "the timed plan's own result (last timed execute) against the CPU oracle on the same noise"}
    # all cores: the channels are independent for this pipeline -> one single-channel sink per thread
    ncpu = os.cpu_count() or 1
    nthr = min(nch, ncpu)
    cols = [np.asfortranarray(noise[:, c:c + 1]) for c in range(nch)]
    trees = [tree_fn(so, c, n_in) for c in cols]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as pool:
        ys = list(pool.map(ob.oracle_sink, trees))
    dtn = time.perf_counter() - t0
    same = all(np.array_equal(ys[c][:, 0], y[:, c]) for c in range(nch))
    return full, {"value": y.shape[0] / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"the whole workload: {seconds:g} s x {nch} ch ({y.shape[0]} output frames), blocksize 4096, 1 thread",
            "seconds": dt1, "flags": flags + " -ffp-contract=off", "cpu_model": cpu_model(), "host_cpus": ncpu,
            "all_cores": {"value": y.shape[0] / dtn, "unit": "frames/s", "cores": nthr, "seconds": dtn,
                          "how": "one single-channel sink per thread (channels are independent)",
                          "identical_to_1_thread": bool(same)}}


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="ns", choices=["ns", "config3", "config4", "config5", "ns_time"])
    ap.add_argument("--seconds", type=float, default=600.0, help="signal duration (600 = full config)")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"])
    ap.add_argument("--cpu-seconds", type=float, default=600.0,
                    help="signal seconds for the CPU-oracle sample (0 = skip); the full 600 s x 8 ch pipeline "
                         "is ~10-20 s of CPU work on one core")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-3 object")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the one-shot latency probe (three child processes)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    import numpy as np
    import torch

    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        ge.build()
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() == 0:
            dist.init_process_group("gloo")
            return launch_path_only(args, torch, dist, rank, world)
        if torch.cuda.device_count() >= world:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            # fewer GPUs than ranks (a one-GPU box): the ranks share devices and synchronise over gloo -- a way to
            # run the N > 1 code path of the default workload where RCCL refuses two ranks on one device; the
            # numbers of such a run mean nothing and say so ("oversubscribed")
            dist.init_process_group("gloo")
            local_rank = local_rank % max(1, torch.cuda.device_count())
        dist.barrier()
    import sigops_amd as so

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the sink engine has no CPU path")
    rccl_ranks = dist.get_world_size() if dist is not None and dist.get_backend() == "nccl" else None
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if args.workload in ("config4", "config5", "ns_time"):
        import bench_multi

        return bench_multi.run(args, so, torch, dist, rank, local_rank, world, dev)
    config4 = None

    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ndt = np.float64 if args.dtype == "f64" else np.float32
    nch = args.channels
    n_in = int(round(args.seconds * 44100))
    stream = torch.cuda.current_stream().cuda_stream

    gen = torch.Generator(device=dev)
    gen.manual_seed(1983 + rank)
    noise_t = torch.randn((nch, n_in), dtype=tdt, device=dev, generator=gen)
    noise = noise_t.t()  # [n_in x nch], column-major like Julia's Array (time fastest)

    def prepare(tree_fn):
        x = tree_fn(so, noise, n_in)
        n_out = so.nframes(x)
        out_t = torch.empty((nch, n_out), dtype=tdt, device=dev)
        out = out_t.t()
        t0 = time.perf_counter()
        plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), ndt, (out.stride(0), out.stride(1)), True, device=local_rank)
        torch.cuda.synchronize()
        return plan, out_t, out, n_out, (time.perf_counter() - t0) * 1e3

    headline_fn = tree_ns if args.workload == "ns" else tree_config3
    headline_name = ("north-star pipeline: Mix(sin 1kHz, noise[%d x %d] @44.1kHz) |> Until(%gs) |> Filt(Bandstop 0.5-2kHz) "
                     "|> ToFramerate(48kHz) |> sink" if args.workload == "ns" else
                     "config3: Signal(noise[%d x %d],44.1kHz) |> Amplify(sin 5Hz) |> Until(%gs) |> ToFramerate(48kHz) |> sink") \
        % (n_in, nch, args.seconds) + " (device-resident leaf and result)"

    plan, out_t, out, n_out, plan_ms = prepare(headline_fn)
    optr = out.data_ptr()
    # (all plans before the first timed loop: plan creation is host work -- the replay of DSP.jl's phase accumulator --
    #  during which the device idles and its power management falls back; the second workload then starts right behind
    #  the first loop instead of behind such a gap)
    do_secondary = rank == 0 and world == 1 and args.workload == "ns" and not args.no_secondary
    if do_secondary:
        p3, o3_t, o3, n3, plan3_ms = prepare(tree_config3)
    series = []
    elapsed = timed_loop(plan, optr, stream, args.steps, args.warmup, torch, dist, dev, series=series, profile=True)
    host_side = series.pop() if series and isinstance(series[-1], dict) else None
    stages = stage_means(plan)  # per-kernel means over the timed executes themselves
    st = plan.stats()
    issue = dict(plan.counters(), profiling="deferred: per-kernel events recorded during the timed executes, never "
                 "synchronising; a plan in this mode issues direct launches (plans of four or more steps replay a HIP "
                 "graph outside the bench -- the headline is one or two launches either way)")
    secondary = None
    if do_secondary:
        # (the secondary workload's own loop, at least 100 / 30 whatever --steps / --warmup are: a 20-step loop right behind
        #  a change of load sits inside the chip's power-management transient -- `step_ms_series` shows it, and
        #  `first_20_ms_per_step` is what such a loop would have reported)
        steps3, warmup3 = max(100, args.steps // 2), max(30, args.warmup)
        series3 = []
        e3 = timed_loop(p3, o3.data_ptr(), stream, steps3, warmup3, torch, None, dev, series=series3, profile=True)
        host3 = series3.pop() if series3 and isinstance(series3[-1], dict) else None
        st3 = stage_means(p3)
        s3 = p3.stats()
        ms3 = e3 / steps3 * 1e3
        dom3 = max(st3, key=lambda s: s["ms"])
        tr3, src3 = pmc_traffic("config3:" + dom3["name"])
        secondary = {"steady_state_ms": (sum(series3[-10:]) / len(series3[-10:])) if series3 else None,
                     "workload": "config3: Signal(noise[%d x %d],44.1kHz) |> Amplify(sin 5Hz) |> Until(%gs) |> ToFramerate(48kHz) |> sink"
                                 % (n_in, nch, args.seconds),
                     "value": n3 / (ms3 * 1e-3), "unit": "frames/s", "steps": steps3, "warmup": warmup3, "ms_per_step": ms3,
                     "first_20_ms_per_step": (sum(series3[5:25]) / len(series3[5:25])) if len(series3) >= 25 else None,
                     "algorithmic_bytes_per_step": s3["algorithmic_bytes"],
                     "roofline_sink": {"achieved": s3["algorithmic_bytes"] / (ms3 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": s3["algorithmic_bytes"] / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "frac_of_copy_ceiling": s3["algorithmic_bytes"] / (ms3 * 1e-3) / 1e9 / HBM_COPY_GBS,
                                       "from": "timed loop"},
                     "roofline": roofline_of(dom3, tr3, src3), "stages": st3, "plan_create_ms": plan3_ms,
                     "step_ms_series": series3, "host_side": host3}
        p3.close()
        del o3_t, o3
    checksum = float(out_t[:, :: max(1, n_out // 4096)].double().abs().sum().item())
    plan.close()
    gpu_result = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        gpu_result = np.asfortranarray(out_t.t().cpu().numpy())  # [n_out x nch], for the full-length comparison
    del out_t, out

    # ---- N > 1 over RCCL: the path that has an exchange step, next to the headline (every rank takes part) ----
    if dist is not None and dist.get_backend() == "nccl" and args.workload == "ns":
        import bench_multi

        a4 = argparse.Namespace(**vars(args))
        a4.workload = "config4"
        a4.steps, a4.warmup = max(5, args.steps // 4), max(2, args.warmup // 2)
        # (the headline line must not depend on the secondary workload -- nor hang on it: bench_multi.run agrees over
        #  all ranks, after its local set-up and after a first local execute, that every rank is still there before it
        #  enters its barriers and all-gathers; a rank that failed makes all of them skip the workload)
        config4 = bench_multi.run(a4, so, torch, dist, rank, local_rank, world, dev, emit=False)

    # ---- correctness gate (rank 0, same device noise, a prefix); the whole length: parity_full below ----
    gate = None
    if rank == 0:
        m = min(n_in, 150_000)
        pre = np.asfortranarray(noise_t[:, :m].t().cpu().numpy())
        gate = parity_gate(so, headline_fn, pre, 1e-6)

    if rank == 0:
        algo = st["algorithmic_bytes"]
        ms_per_step = elapsed / args.steps * 1e3
        dom = max(stages, key=lambda s: s["ms"])
        # (the PMC file's tags: the workload, "_f32" for a Float32 leaf and result, "_<n>ch" for another channel count --
        #  tools/collect_r06.sh)
        pmc_tag = args.workload + ("_f32" if args.dtype == "f32" else "") + ("_%dch" % args.channels if args.workload == "ns" and args.channels != 8 else "")
        traffic, tsrc = pmc_traffic(pmc_tag + ":" + dom["name"])
        sink_gbps = algo / (ms_per_step * 1e-3) / 1e9
        res = {
            "metric": METRIC,
            "value": world * n_out * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": headline_name, "in_frames": n_in, "out_frames": n_out, "channels": nch,
                       "parallelism": f"{world} independent signals, one per GPU, no collective"
                                      + (" (OVERSUBSCRIBED: more ranks than GPUs, gloo barriers; not a measurement)"
                                         if dist is not None and dist.get_backend() != "nccl" else ""),
                       "plan_create_ms": plan_ms, "launches_per_step": st["n_launches"], "stages": st["n_stages"],
                       "scratch_bytes": st["scratch_bytes"], "checksum": checksum,
                       "note": "warm-up is exactly --warmup steps; the first ~15 launches after an idle gap run up to 20% slower"},
            "algorithmic_bytes_per_step": algo,
            "roofline": roofline_of(dom, traffic, tsrc),
            "roofline_mfma": mfma_roofline_of(dom, n_out, nch, issue.get("fused_mfmas_per_block")),
            "roofline_sink": {"bound": "hbm", "achieved": sink_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": sink_gbps / HBM_PEAK_GBS, "frac_of_copy_ceiling": sink_gbps / HBM_COPY_GBS,
                              "from": "timed loop: algorithmic bytes of the sink (leaf read + result written) / ms_per_step"},
            "stages": stages,
            "issue": issue,
            "steady_state_ms": (sum(series[-10:]) / len(series[-10:])) if series else None,
            "steady_state_note": "mean device time of the last 10 executes of step_ms_series (the kernel once the chip's "
                                 "power management has settled); ms_per_step is the whole timed loop on the host clock",
            "rccl_ranks": rccl_ranks,
            "step_ms_series": series,
            "host_side": host_side,
            "step_ms_series_note": "device time of every execute, the --warmup ones first (events on the launch stream, no "
                                   "synchronisation between executes).  After a load step MI355X's power management runs "
                                   "executes 3-15 up to 40 % slower and settles after ~30 (DESIGN.md, cold-run study): "
                                   "short runs (20 / 5) sit inside that transient, long ones (200 / 30) mostly outside.",
            "parity_gate": gate,
            "config3": secondary,
            "config4": config4,
        }
        if world == 1 and args.workload == "ns" and not args.no_one_shot and not args.no_secondary:
            res["one_shot"] = one_shot_probe()
        if args.cpu_seconds > 0 and world == 1:
            noise_host = np.asfortranarray(noise_t.t().cpu().numpy())
            full, res["cpu_baseline"] = cpu_baseline(so, headline_fn, args.cpu_seconds, nch, ndt, noise_host, gpu_result)
            res["parity_full"] = full
            tol = 1e-6
            if full is not None and not (full["relerr"] <= tol):
                raise SystemExit(f"bench.py: the timed result differs from the CPU oracle: {full}")
        else:
            res["cpu_baseline"] = None
            res["parity_full"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
