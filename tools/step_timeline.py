#!/usr/bin/env python
"""Timeline analysis of one training step from a rocprofv3 kernel trace of bench.py: per hardware queue and per kernel class the busy
time, the union of all busy intervals (wall time with >= 1 kernel running), idle gaps, and how many kernels run concurrently.
A step = the interval between two consecutive adamw_kernel launches (the last complete one in the trace)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows), key=lambda e: e[0])
adam = [e[0] for e in ev if "adamw_kernel" in e[2]]
big = [t for i, t in enumerate(adam) if i == 0 or t - adam[i - 1] > 5_000_000]     # first adamw launch of each step (151 M params: one big launch)
t0, t1 = big[-2], big[-1]
if len(sys.argv) > 2 and sys.argv[2] == "longest":     # the step with the most dispatches (bench.py's default run: the global-batch step, followed by shorter variants)
    import bisect
    starts = [e[0] for e in ev]
    k = max(range(len(big) - 1), key=lambda i: bisect.bisect_left(starts, big[i + 1]) - bisect.bisect_left(starts, big[i]))   # most dispatches
    t0, t1 = big[k], big[k + 1]
step = [e for e in ev if t0 <= e[0] < t1]
print(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernel dispatches")
def cls(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("gemm_bf16_nt", "gemm_bf16_tn", "splitk_reduce", "attn_bwd", "attn_fwd", "layernorm_bwd", "layernorm_fwd", "partial_reduce", "transpose_cast", "adamw",
              "gemm_f32", "colsum"):
        if k in n:
            return k
    return "other"
by = collections.defaultdict(float)
for s, e, n, q in step:
    by[cls(n)] += (e - s) / 1e6
print("sum of kernel durations per class (ms):", {k: round(v, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}, "total", round(sum(by.values()), 2))
# union and concurrency histogram
pts = sorted([(s, 1) for s, e, n, q in step] + [(e, -1) for s, e, n, q in step])
cur, last, hist = 0, t0, collections.defaultdict(float)
for t, d in pts:
    hist[min(cur, 6)] += (t - last) / 1e6
    cur += d
    last = t
print("time with k kernels in flight (ms):", {k: round(v, 2) for k, v in sorted(hist.items())})
# time attributed exclusively: when exactly one class of GEMM is running etc.
pts2 = sorted([(s, 1, cls(n)) for s, e, n, q in step] + [(e, -1, cls(n)) for s, e, n, q in step], key=lambda x: (x[0], x[1]))
active, last, gemm_busy = collections.Counter(), t0, 0.0
for t, d, c in pts2:
    if active["gemm_bf16_nt"] + active["gemm_bf16_tn"] > 0:
        gemm_busy += (t - last) / 1e6
    active[c] += d
    last = t
print(f"a bf16 GEMM kernel is running during {gemm_busy:.2f} ms of the step")
by_q = collections.defaultdict(float)
for s, e, n, q in step:
    by_q[q] += (e - s) / 1e6
print("busy time per hardware queue (ms):", {k: round(v, 2) for k, v in sorted(by_q.items())})
# what runs while NO bf16 GEMM is in flight: exposed time per kernel class (a moment with several such kernels is split between them)
active2, last, exposed = collections.Counter(), t0, collections.defaultdict(float)
names = collections.defaultdict(collections.Counter)
live = []
for t, d, c in pts2:
    if active2["gemm_bf16_nt"] + active2["gemm_bf16_tn"] == 0 and t > last:
        others = [k for k, v in active2.items() if v > 0]
        if others:
            for k in others:
                exposed[k] += (t - last) / 1e6 / len(others)
        else:
            exposed["(nothing running)"] += (t - last) / 1e6
    active2[c] += d
    last = t
print("time with no bf16 GEMM in flight, by what runs instead (ms):", {k: round(v, 2) for k, v in sorted(exposed.items(), key=lambda kv: -kv[1])})
other = collections.defaultdict(float)
for s, e, n, q in step:
    if cls(n) == "other":
        other[n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]] += (e - s) / 1e6
print("'other' kernels by summed duration (ms):", {k: round(v, 2) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:14]})
