#!/usr/bin/env python3
"""Does the exchange run UNDER the interior rows' product?  Reads a rocprofv3 --kernel-trace CSV of a partitioned run (several
blocks on one device, PEER backend: the exchange is the pull kernel k_peer_pull) and reports, for every pull kernel, how much of
its duration overlapped in time with SpMV kernels of other streams.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --gpus 4 ...   (SMH_BENCH_SHARE_DEVICES=1)
    python3 tools/overlap_trace.py DIR
"""
import csv
import glob
import re
import sys


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit("no kernel trace under " + sys.argv[1])
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            m = re.search(r"\bk_\w+(<[^(]*>)?", r["Kernel_Name"])  # ("(anonymous namespace)::k_peer_pull(...)" starts with a parenthesis)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:60],
                         r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    spmv = [r for r in rows if "k_spmv" in r[2]]
    pulls = [r for r in rows if "k_peer_pull" in r[2]]
    print("kernels: %d in the trace, %d SpMV launches, %d pull launches" % (len(rows), len(spmv), len(pulls)))
    if not pulls:
        return
    hidden, total, under = 0, 0, 0
    j = 0
    for (a, e, _, q) in pulls:
        total += e - a
        while j < len(spmv) and spmv[j][1] < a:
            j += 1
        ov, k = 0, j
        while k < len(spmv) and spmv[k][0] < e:
            if spmv[k][3] != q:
                ov = max(ov, min(e, spmv[k][1]) - max(a, spmv[k][0]))
            k += 1
        ov = max(0, ov)
        hidden += ov
        under += ov > 0
    print("pull kernels that ran while an SpMV kernel of another stream was running: %d of %d" % (under, len(pulls)))
    print("pull time under a concurrent SpMV: %.1f us of %.1f us (%.0f %%)" % (hidden / 1e3, total / 1e3, 100.0 * hidden / max(1, total)))
    # a sample of the timeline: the first pull with overlap and the kernels around it
    for (a, e, name, q) in pulls:
        near = [r for r in rows if r[1] > a - 50_000 and r[0] < e + 50_000]
        if any("k_spmv" in r[2] and r[3] != q and r[0] < e and r[1] > a for r in near):
            t0 = near[0][0]
            print("sample (us relative to the first kernel shown; stream / queue id in brackets):")
            for r in near[:14]:
                print("  %9.1f .. %9.1f  [%s]  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2][:70]))
            break


if __name__ == "__main__":
    main()
