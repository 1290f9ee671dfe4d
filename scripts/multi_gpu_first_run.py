#!/usr/bin/env python3
"""The FIRST multi-GPU run, as one command: nothing in this repository has ever executed on two GPUs -- the build
boxes have one, the driver's 8-GPU node has been unavailable in every round -- so the first node that has them should yield numbers,
not a traceback.  Runs, in order, each as its own process group / process with a hard timeout, and writes ONE JSON object per phase
(a result, or {"error": ..., "tail": ...}) to the output file and to stdout:

  1. bench.py --gpus N                                  configs[1] replicated, gradient merge tuned on the node (StepLoop.tune_merge)
  2. bench.py --gpus N --config C4 --parallelism sharded-local        configs[3], entity table row-sharded, shard-local negatives
  3. bench.py --gpus N --config C5 --ents-per-gpu E ...  a cut-down configs[4] (E rows per GPU; the full 6.25 M with --full-c5)
  4. bench.py --gpus N --parallelism columns            configs[1] column-sharded (the measured-negative design, for the record)
  5. the C-ABI session group, numpy only (no torch in that process): SessionGroup(devices=range(N)) replicated train steps,
     SessionGroup(rows=True) train steps + rank() (library-owned RCCL: ncclCommInitAll, grouped ncclAllReduce / ncclSend / ncclRecv,
     one host thread per device in rank()), each phase timed.

usage:  python scripts/multi_gpu_first_run.py [--gpus 8] [--out multi_gpu_first_run.jsonl] [--timeout 600] [--full-c5]
Dry run on ONE GPU (what profiles/r06_multi_gpu_dry_run_gloo8.jsonl is):
        AMDKGE_BENCH_BACKEND=gloo python scripts/multi_gpu_first_run.py --gpus 8 --same-device
  (--same-device: the torch phases run their N ranks on the one GPU over gloo, the session-group phase puts its N replicas on
  device 0 -- the exchanges are then host-staged / local kernels and say nothing about xGMI; the point is that every phase completes.)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_phase(name, cmd, timeout, env, out_f, parse="last_json"):
    t0 = time.time()
    rec = {"phase": name, "cmd": " ".join(cmd)}
    try:
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        rec["rc"], rec["wall_s"] = p.returncode, round(time.time() - t0, 1)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            rec["result"] = json.loads(lines[-1])
        else:
            rec["error"] = f"rc {p.returncode}, no JSON line" if p.returncode == 0 else f"rc {p.returncode}"
            rec["tail"] = (p.stderr or p.stdout)[-1500:]
    except subprocess.TimeoutExpired as e:
        rec["error"], rec["wall_s"] = f"timeout after {timeout} s", round(time.time() - t0, 1)
        rec["tail"] = ((e.stderr or b"")[-1500:].decode("utf-8", "replace") if isinstance(e.stderr, bytes) else str(e.stderr or "")[-1500:])
    except Exception as e:   # noqa: BLE001 -- a phase never costs the next one
        rec["error"], rec["wall_s"] = f"{type(e).__name__}: {e}", round(time.time() - t0, 1)
    out_f.write(json.dumps(rec) + "\n")
    out_f.flush()
    brief = {k: rec.get(k) for k in ("phase", "rc", "wall_s", "error")}
    if "result" in rec and isinstance(rec["result"], dict):
        r = rec["result"]
        brief.update({k: r.get(k) for k in ("value", "unit", "ms_per_step", "n_gpus") if k in r})
        if "phases_ms" in r:
            brief["phases_ms"] = r["phases_ms"]
        if "phases_s" in r:
            brief["phases_s"] = r["phases_s"]
    print(json.dumps(brief), flush=True)
    return rec


SESSION_GROUP = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from ampligraph_amd.latent_features import loss_functions, optimizers
from ampligraph_amd.session import Session, SessionGroup
N_GPUS, SAME = %(gpus)d, %(same)d
devices = [0] * N_GPUS if SAME else list(range(N_GPUS))
rng = np.random.default_rng(0)
out = {"devices": devices, "phases_s": {}}
def clock(name, fn):
    t0 = time.perf_counter(); r = fn(); out["phases_s"][name] = round(time.perf_counter() - t0, 4); return r
# ---- replicated group: configs[1]'s model on a cut-down graph, data-parallel steps through library-owned RCCL ----
N, R, k, eta, B = 14505, 237, 200, 20, 10000 * N_GPUS
mk = lambda: (loss_functions.get("self_adversarial"), optimizers.get("adam"))
K = 2 * k
ent = (rng.uniform(-1, 1, size=(N, K)) * 0.02).astype(np.float32); rel = (rng.uniform(-1, 1, size=(R, K)) * 0.1).astype(np.float32)
X = np.stack([rng.integers(0, N, 4 * B), rng.integers(0, R, 4 * B), rng.integers(0, N, 4 * B)], 1).astype(np.int32)
g = clock("replicated_create", lambda: SessionGroup(devices, "ComplEx", k, N, R, eta, *mk(), seed=0))
out["replicated_uses_rccl"], out["rccl_version"] = g.info()
clock("replicated_set_rows", lambda: (g.set_rows("ent", ent), g.set_rows("rel", rel)))
losses = [clock(f"replicated_train_step_{i}", lambda i=i: g.train_step(X[i * B:(i + 1) * B])) for i in range(4)]
out["replicated_losses"] = [float(l) for l in losses]
T = X[:2048]
clock("replicated_rank_2048_queries", lambda: g.rank(T, corrupt_side="s,o"))
g.close()
# ---- row-sharded group: configs[3]'s shape (123 182 entities), shard-local negatives, then evaluation through the group ----
N4, B4 = 123182, 8192 * N_GPUS
ent4 = (rng.uniform(-1, 1, size=(N4, K)) * 0.01).astype(np.float32)
X4 = np.stack([rng.integers(0, N4, 3 * B4), rng.integers(0, R, 3 * B4), rng.integers(0, N4, 3 * B4)], 1).astype(np.int32)
gr = clock("rows_create", lambda: SessionGroup(devices, "ComplEx", k, N4, R, eta, *mk(), seed=0, rows=True, max_batch=B4))
clock("rows_set_rows", lambda: (gr.set_rows("ent", ent4), gr.set_rows("rel", rel)))
out["rows_losses"] = [float(clock(f"rows_train_step_{i}", lambda i=i: gr.train_step(X4[i * B4:(i + 1) * B4]))) for i in range(3)]
T4 = X4[:4096]
ranks = clock("rows_rank_4096_queries", lambda: gr.rank(T4, corrupt_side="s,o"))
out["rows_rank_mrr"] = float(np.mean(1.0 / ranks))
one = Session("ComplEx", k, N4, R, eta, *mk(), seed=0)
one.set_rows("ent", gr.get_rows("ent")); one.set_rows("rel", gr.get_rows("rel"))
out["rows_rank_equals_one_session"] = bool(np.array_equal(ranks, one.rank(T4, corrupt_side="s,o")))
one.close(); gr.close()
print(json.dumps(out))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--out", default="multi_gpu_first_run.jsonl")
    ap.add_argument("--timeout", type=int, default=600, help="hard limit per phase, seconds")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--full-c5", action="store_true", help="the whole 6.25 M rows per GPU of configs[4] (200 GB resident per GPU)")
    ap.add_argument("--same-device", action="store_true", help="dry run on one GPU: N ranks / replicas on device 0 (set AMDKGE_BENCH_BACKEND=gloo)")
    a = ap.parse_args()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n = str(a.gpus)
    common = ["--no-cpu-baseline", "--no-eval", "--also", "none", "--steps", str(a.steps), "--warmup", "5", "--reps", "3"]
    c5 = ["--config", "C5", "--steps", "6", "--warmup", "2", "--reps", "2", "--no-cpu-baseline", "--no-eval", "--also", "none"]
    if not a.full_c5:
        c5 += ["--ents-per-gpu", "500000", "--batch", "16384"]
    py = sys.executable
    with open(a.out, "a") as f:
        f.write(json.dumps({"phase": "start", "gpus": a.gpus, "backend": env.get("AMDKGE_BENCH_BACKEND", "nccl"), "same_device": a.same_device,
                            "time": time.strftime("%Y-%m-%d %H:%M:%S")}) + "\n")
        run_phase("C2 replicated, merge tuned on the node", [py, "bench.py", "--gpus", n] + common, a.timeout, env, f)
        run_phase("C4 row-sharded, shard-local negatives", [py, "bench.py", "--gpus", n, "--config", "C4", "--parallelism", "sharded-local"] + common, a.timeout, env, f)
        run_phase("C5 row-sharded" + ("" if a.full_c5 else " (cut down: 500 000 rows per GPU, B = 16 384)"), [py, "bench.py", "--gpus", n] + c5, a.timeout, env, f)
        if 200 % a.gpus == 0:
            run_phase("C2 column-sharded", [py, "bench.py", "--gpus", n, "--parallelism", "columns"] + common, a.timeout, env, f)
        run_phase("C-ABI session groups (numpy host, library-owned RCCL): replicated + row-sharded train, rank",
                  [py, "-c", SESSION_GROUP % {"root": ROOT, "gpus": a.gpus, "same": 1 if a.same_device else 0}], a.timeout, env, f)


if __name__ == "__main__":
    main()
