"""Per-step, per-kernel totals from a minimal kernel trace (Kernel_Name, Queue_Id, Start_Timestamp, End_Timestamp, as rocprofv3 --kernel-trace writes them): a step ends with the
one k_voxel_centroids launch of updateScansScanwise.  Profiling helper."""
import collections
import csv
import gzip
import sys


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    for r in csv.DictReader(op(path, "rt")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].replace("ltm::", "").split("<")[0]))
    rows.sort()
    last = max((i for i, r in enumerate(rows) if "k_selfcheck" in r[3]), default=-1)
    return rows[last + 1:]


def steps_of(rows):
    ends = [e for s, e, q, n in rows if n == "k_voxel_centroids"]
    out, lo = [], rows[0][0]
    for e in ends:
        out.append([r for r in rows if lo <= r[0] <= e])
        lo = e + 1
    return out


def main():
    rows = load(sys.argv[1])
    st = steps_of(rows)
    tots = []
    for i, rs in enumerate(st):
        t = collections.defaultdict(lambda: [0, 0])
        for s, e, q, n in rs:
            t[n][0] += e - s
            t[n][1] += 1
        tots.append(t)
        qs = sorted(set(r[2] for r in rs))
        print(f"step {i}: {len(rs)} launches on queues {qs}, span {(max(r[1] for r in rs) - rs[0][0]) / 1e6:.2f} ms, sum of kernel time {sum(v[0] for v in t.values()) / 1e6:.2f} ms")
    if len(tots) >= 2:
        cols = list(range(len(tots)))
        names = sorted(set().union(*[set(t) for t in tots]), key=lambda n: -max(t.get(n, [0, 0])[0] for t in tots))
        print("%-34s" % "kernel" + "".join(f"   step{c} ms (n)" for c in cols))
        for n in names[:34]:
            print("%-34s" % n + "".join("  %8.2f (%3d)" % (tots[c].get(n, [0, 0])[0] / 1e6, tots[c].get(n, [0, 0])[1]) for c in cols))


if __name__ == "__main__":
    main()
