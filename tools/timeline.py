#!/usr/bin/env python3
"""Where the device's time goes between launches: reads a rocprofv3 --kernel-trace CSV and prints, for the steady part of the run,
per-kernel durations, the gap in front of each launch on its own queue, how much of the wall time has 0 / 1 / 2+ kernels running,
and the frame period.  usage: python tools/timeline.py <dir or kernel_trace.csv> [skip_fraction]"""
import csv, glob, os, sys
from collections import defaultdict

def short(name):
    for k in ("k_ingest_wave", "k_build_wave_weights", "k_yuv420_to_rgba", "k_yuv_to_rgba_batch", "k_ingest_resample", "k_compose_output",
              "k_classify_tiles", "k_apply_layouts", "k_blit_glyphs"):
        if k in name: return k
    return name[:40]

ALL = os.environ.get("TL_ALL") == "1"          # every kernel of the trace, not only the three hot ones (configs[4]: shader, classify, text)
SAMPLE = int(os.environ.get("TL_SAMPLE", "12"))  # launches printed with their relative times

def main():
    p = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
    for f in files:
        rows = []
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
        rows.sort()
        if not rows: continue
        hot = [r for r in rows if ALL or r[3] in ("k_ingest_wave", "k_yuv420_to_rgba", "k_compose_output")]
        if not hot: continue
        # the longest stretch without a gap above 1 ms = one timed loop; keep its last (1 - skip)
        runs, cur = [], [hot[0]]
        for r in hot[1:]:
            if r[0] - cur[-1][1] > 1_000_000: runs.append(cur); cur = [r]
            else: cur.append(r)
        runs.append(cur)
        run = max(runs, key=len)
        run = run[int(len(run) * skip):]
        t0, t1 = run[0][0], max(r[1] for r in run)
        print(f"== {os.path.relpath(f)}: {len(run)} launches over {(t1 - t0) / 1e3:.1f} us ({len(runs)} stretches; longest kept, first {skip:.0%} skipped)")
        dur = defaultdict(list)
        for s, e, q, k in run: dur[k].append(e - s)
        for k, v in dur.items():
            print(f"   {k:22s} n {len(v):5d}  avg {sum(v) / len(v) / 1e3:7.2f} us  min {min(v) / 1e3:7.2f}  max {max(v) / 1e3:7.2f}")
        nfr = len(dur.get("k_compose_output", [])) or 1
        print(f"   frame period {(t1 - t0) / nfr / 1e3:.2f} us  ({nfr} frames)  sum of kernel averages {sum(sum(v) / len(v) for v in dur.values()) / 1e3:.2f} us")
        # gaps per queue
        lastq, gaps = {}, defaultdict(list)
        for s, e, q, k in run:
            if q in lastq: gaps[k].append(s - lastq[q])
            lastq[q] = e
        for k, v in gaps.items():
            print(f"   gap before {k:22s} on its queue: avg {sum(v) / len(v) / 1e3:7.2f} us  min {min(v) / 1e3:7.2f}  max {max(v) / 1e3:7.2f}")
        # concurrency
        ev = []
        for s, e, q, k in run: ev += [(s, 1), (e, -1)]
        ev.sort()
        lvl, last, occ = 0, t0, defaultdict(int)
        for t, d in ev:
            occ[min(lvl, 2)] += t - last; last = t; lvl += d
        tot = sum(occ.values()) or 1
        print("   wall time with 0 / 1 / 2+ kernels running: " + " / ".join(f"{100.0 * occ[i] / tot:.1f}%" for i in range(3)))
        # pairwise overlap matrix
        ov = defaultdict(int)
        active = []
        for s, e, q, k in run:
            active = [a for a in active if a[1] > s]
            for a in active: ov[tuple(sorted((a[3], k)))] += min(a[1], e) - s
            active.append((s, e, q, k))
        for (a, b), v in sorted(ov.items(), key=lambda x: -x[1]):
            print(f"   overlap {a} x {b}: {v / nfr / 1e3:.2f} us per frame")
        # one frame's worth of launches, relative times
        print("   sample (us from the first):")
        for s, e, q, k in run[:SAMPLE]:
            print(f"      q{q} {k:22s} {(s - run[0][0]) / 1e3:8.2f} -> {(e - run[0][0]) / 1e3:8.2f}")

if __name__ == "__main__":
    main()
