"""Condense a rocprofv3 --kernel-trace CSV into a readable timeline of one step: which queue ran what, when, and how busy the device was.

    python tools/trace_timeline.py <kernel_trace.csv> [--from-kernel NAME] [--bin-us 1000] [--top 40]

Prints (a) per queue: busy time, launches; (b) a binned timeline: per bin of --bin-us the fraction of the bin each queue had a kernel in flight
and the dominant kernel; (c) the idle gaps (no kernel on any queue) longer than 20 us, summed.  Test / profiling helper, not product code.
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"^void\s+", "", name)
    if "rocprim" in name:
        for a in ("radix_sort_onesweep", "radix_sort_histogram", "radix_sort_block_sort", "merge_sort_block_merge", "merge_sort_block_sort", "lookback_scan_state", "scan", "transform", "partition", "select"):
            if a in name:
                return "rocprim::" + a
        return "rocprim::other"
    return name.split("(")[0].replace("ltm::", "").split("<")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--bin-us", type=float, default=1000.0)
    ap.add_argument("--skip-before", default="k_selfcheck", help="drop everything up to the LAST launch of this kernel (context creation)")
    ap.add_argument("--max-bins", type=int, default=400)
    ap.add_argument("--window", action="append", default=[], help="a,b [ms from the first launch]: per-kernel totals inside the window (repeatable)")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        n = r["Kernel_Name"]
        if "at::native" in n or "ROCPRIM_400001" in n or "rocclr" in n or "Cijk_" in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(n)))
    rows.sort()
    last = max((i for i, r in enumerate(rows) if a.skip_before in r[3]), default=-1)
    rows = rows[last + 1:]
    if not rows:
        print("no kernels")
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"{len(rows)} launches, span {(t1 - t0) / 1e6:.2f} ms")
    per_q = collections.defaultdict(lambda: [0, 0])
    for s, e, q, n in rows:
        per_q[q][0] += e - s
        per_q[q][1] += 1
    for q, (busy, cnt) in sorted(per_q.items()):
        print(f"  queue {q}: {cnt} launches, {busy / 1e6:.2f} ms of kernel time")
    # union busy / idle gaps
    ev = sorted((s, e) for s, e, _, _ in rows)
    cur_s, cur_e = ev[0]
    busy, gaps = 0, []
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    big = [g for g in gaps if g[0] > 20000]
    print(f"device busy (any queue) {busy / 1e6:.2f} ms; idle {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps, {sum(g[0] for g in big) / 1e6:.2f} ms in {len(big)} gaps > 20 us")
    # time with >= 2 queues in flight
    pts = []
    for s, e, q, n in rows:
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    depth, prev, over = 0, pts[0][0], 0
    for t, d in pts:
        if depth >= 2:
            over += t - prev
        depth += d; prev = t
    print(f"time with >= 2 kernels in flight: {over / 1e6:.2f} ms")
    for w in a.window:
        lo, hi = (float(x) * 1e6 + t0 for x in w.split(","))
        tot = collections.defaultdict(lambda: [0, 0])
        for s, e, q, n in rows:
            if s >= lo and s < hi:
                tot[n][0] += e - s
                tot[n][1] += 1
        print(f"window {w} ms: {sum(v[1] for v in tot.values())} launches, {sum(v[0] for v in tot.values()) / 1e6:.2f} ms of kernel time")
        for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:30]:
            print(f"    {d / 1e6:8.3f} ms  {c:5d}  {n}")
    nb = int((t1 - t0) / (a.bin_us * 1000)) + 1
    if nb > a.max_bins:
        print(f"({nb} bins; showing the first {a.max_bins})")
        nb = a.max_bins
    qs = sorted(per_q)
    bins = [collections.defaultdict(lambda: collections.defaultdict(int)) for _ in range(nb)]
    w = a.bin_us * 1000
    for s, e, q, n in rows:
        b0, b1 = int((s - t0) / w), int((e - t0) / w)
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * w), min(e, t0 + (b + 1) * w)
            if hi > lo:
                bins[b][q][n] += hi - lo
    print("bin[ms]  " + "  ".join(f"queue {q}: busy% dominant kernel".ljust(44) for q in qs))
    for b in range(nb):
        cells = []
        for q in qs:
            d = bins[b][q]
            tot = sum(d.values())
            dom = max(d, key=d.get) if d else "-"
            cells.append(f"{100 * tot / w:5.0f}% {dom}".ljust(44))
        print(f"{b * a.bin_us / 1000:7.1f}  " + "  ".join(cells))


if __name__ == "__main__":
    main()
