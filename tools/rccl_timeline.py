#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of `SNWS_DP_SELFTEST=1 python3 bench.py ...` (a ONE-rank RCCL group on a
one-GPU box: every collective of the data-parallel path is issued, none moves data between GPUs): how many RCCL
kernels a training step launches, on which queue / stream, how long they run, whether the gradient-bucket all-reduces
overlap backward kernels, and what a statistics all-reduce costs the compute stream (the idle window between the
kernel before it and the kernel after it, against the ordinary launch gap).

    python3 tools/rccl_timeline.py <kernel_trace.csv> [steps to skip at the front]
"""
import csv
import statistics
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    for sep in ("<", "("):
        if sep in name:
            name = name[:name.index(sep)]
    return name.split("::")[-1].split(" ")[-1][:34]


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append(dict(name=r["Kernel_Name"], s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]),
                             q=r.get("Queue_Id", "?"), st=r.get("Stream_Id", "?")))
    rows.sort(key=lambda r: r["s"])
    # the compute stream is the one with the most dispatches; everything RCCL enqueues runs on its communicators' own
    # streams.  With ONE rank RCCL launches no ncclDevKernel at all: an all-reduce becomes two flag fills and one
    # buffer copy (__amd_rocclr_fillBufferAligned / copyBuffer) on the communicator's stream - those are counted here.
    per_stream = {}
    for r in rows:
        per_stream[r["st"]] = per_stream.get(r["st"], 0) + 1
    main_stream = max(per_stream, key=per_stream.get)
    for r in rows:
        r["coll"] = ("nccl" in r["name"].lower() or "rccl" in r["name"].lower() or
                     (r["st"] != main_stream and per_stream[r["st"]] > 8))
    is_rccl = None
    # a training step ends with the fused optimizer kernel
    ends = [i for i, r in enumerate(rows) if "adamw" in r["name"].lower()]
    if len(ends) < skip + 1:
        print("only %d optimizer launches in the trace" % len(ends))
        return
    print("trace: %d kernel dispatches, %d optimizer steps; steps analysed: %d (the first %d are warm-up)"
          % (len(rows), len(ends), len(ends) - skip, skip))
    queues = {}
    for r in rows:
        k = (r["q"], r["st"], r["coll"])
        queues[k] = queues.get(k, 0) + 1
    for (q, st, rc), n in sorted(queues.items()):
        print("  queue %s stream %s: %6d %s dispatches" % (q, st, n, "collective-stream" if rc else "compute"))
    for si in range(skip, len(ends)):
        lo, hi = ends[si - 1] + 1, ends[si] + 1
        step = rows[lo:hi]
        t0, t1 = step[0]["s"], step[-1]["e"]
        comp = [r for r in step if not r["coll"]]
        coll = [r for r in step if r["coll"]]
        busy = sum(r["e"] - r["s"] for r in comp)
        print("step %d: %.3f ms wall, %d compute kernels (%.3f ms busy), %d dispatches on the collective streams (%.3f ms)"
              % (si, (t1 - t0) / 1e6, len(comp), busy / 1e6, len(coll), sum(r["e"] - r["s"] for r in coll) / 1e6))
        # where the compute stream idles: gaps between consecutive compute kernels, by (kernel before, kernel after)
        by_pair = {}
        for a_, b_ in zip(comp, comp[1:]):
            by_pair.setdefault((short(a_["name"]), short(b_["name"])), []).append((b_["s"] - a_["e"]) / 1e3)
        total_gap = sum(max(0.0, g) for v in by_pair.values() for g in v)
        print("    idle time of the compute stream inside the step: %.1f us; largest contributors:" % total_gap)
        for (ka, kb), v in sorted(by_pair.items(), key=lambda kv: -sum(kv[1]))[:5]:
            print("      %-34s -> %-30s n = %3d  median %.2f us  sum %.1f us" % (ka, kb, len(v), statistics.median(v), sum(v)))
        if not coll:
            continue
        names = {}
        for r in coll:
            names.setdefault(r["name"][:90], []).append((r["e"] - r["s"]) / 1e3)
        for n, d in names.items():
            print("    %-90s x%3d  median %.2f us  max %.2f us" % (n, len(d), statistics.median(d), max(d)))
        # one collective = the run of dispatches on one collective stream that are less than 20 us apart
        merged = []
        for r in sorted(coll, key=lambda r: (r["st"], r["s"])):
            if merged and merged[-1]["st"] == r["st"] and r["s"] - merged[-1]["e"] < 20000:
                merged[-1]["e"] = max(merged[-1]["e"], r["e"])
                merged[-1]["k"] += 1
            else:
                merged.append(dict(s=r["s"], e=r["e"], st=r["st"], k=1))
        per = {}
        for m in merged:
            per[m["st"]] = per.get(m["st"], 0) + 1
        print("    grouped into %d collectives: %s" % (len(merged), ", ".join("%d on stream %s" % (v, k) for k, v in sorted(per.items()))))
        coll = sorted(merged, key=lambda r: r["s"])
        # overlap of each collective with compute kernels (any queue)
        small, big = [], []
        for r in coll:
            ov = sum(max(0, min(r["e"], c["e"]) - max(r["s"], c["s"])) for c in comp if c["e"] > r["s"] and c["s"] < r["e"])
            prev = max((c["e"] for c in comp if c["e"] <= r["s"]), default=None)
            nxt = min((c["s"] for c in comp if c["s"] >= r["e"]), default=None)
            window = (nxt - prev) / 1e3 if prev is not None and nxt is not None else float("nan")
            (small if ov == 0 else big).append(((r["e"] - r["s"]) / 1e3, ov / 1e3, window))
        gaps = [(b["s"] - a["e"]) / 1e3 for a, b in zip(comp, comp[1:])]
        gaps_pos = [g for g in gaps if g > 0]
        print("    compute-to-compute launch gaps in this step: median %.2f us (n = %d)"
              % (statistics.median(gaps_pos) if gaps_pos else float("nan"), len(gaps_pos)))
        if small:
            w = [x[2] for x in small]
            print("    %d collectives with NO compute kernel running beside them (on the critical path): kernel median %.2f us; "
                  "idle window of the compute stream around each: median %.2f us, sum %.3f ms per step"
                  % (len(small), statistics.median(x[0] for x in small), statistics.median(w), sum(w) / 1e3))
        if big:
            print("    %d collectives overlapped by compute kernels: kernel median %.2f us, overlapped share %.0f %%"
                  % (len(big), statistics.median(x[0] for x in big),
                     100.0 * sum(x[1] for x in big) / max(sum(x[0] for x in big), 1e-9)))


if __name__ == "__main__":
    main()
