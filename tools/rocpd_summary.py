#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs into the text tables kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/kt/bench_results.db            # kernel-trace stats
    python tools/rocpd_summary.py --pmc gpurun_out/prof/pmc_fetch/bench_results.db ...
"""
import sqlite3
import sys


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), max(vgpr_count), "
        "max(sgpr_count), max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name order by 6 desc"
    ).fetchall()
    tot = float(sum(r[5] for r in rows))
    print("# rocprofv3 --kernel-trace --stats  (%s)" % path)
    print("%-58s %6s %12s %12s %12s %7s %5s %5s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr",
                                                          "sgpr", "grid"))
    for r in rows:
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        print("%-58s %6d %12.1f %12.1f %12.1f %6.2f%% %5d %5d %10s" % (name[:58], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                   100.0 * r[5] / tot, r[6], r[7],
                                                                   "%dx%d/%d" % (r[8], r[9], r[10])))


def pmc_stats(paths):
    print("# rocprofv3 --pmc ... --kernel-trace : per-launch averages (counter values are summed over XCDs/SEs)")
    print("%-40s %-22s %16s %6s" % ("kernel", "counter", "avg_per_launch", "n"))
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                           "group by kernel_name, counter_name order by 1, 2").fetchall()
        for r in rows:
            name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if r[3] >= 2:
                print("%-40s %-22s %16.6g %6d" % (name[:40], r[1], r[2], r[3]))


def gap_stats(path, top=25):
    """Idle time of the device between consecutive kernels (start of one minus end of the previous), grouped by the pair."""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    pairs, busy, idle = {}, 0, 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        gap = s1 - e0
        busy += e0 - s0
        if gap > 200000:  # > 0.2 ms: host-side phases (setup, read-backs), not part of the steady state
            continue
        idle += max(gap, 0)
        key = (short(n0), short(n1))
        g = pairs.setdefault(key, [0, 0])
        g[0] += 1
        g[1] += max(gap, 0)
    print("# gaps between consecutive kernels (%s): busy %.1f ms, idle %.1f ms (gaps > 0.2 ms ignored)" % (path, busy / 1e6, idle / 1e6))
    print("%-34s -> %-34s %6s %10s %10s" % ("previous kernel", "next kernel", "n", "avg_gap_us", "total_ms"))
    for (a, b), (n, tot) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-34s -> %-34s %6d %10.1f %10.2f" % (a[:34], b[:34], n, tot / n / 1e3, tot / 1e6))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        pmc_stats(sys.argv[2:])
    elif sys.argv[1] == "--gaps":
        gap_stats(sys.argv[2])
    else:
        kernel_stats(sys.argv[1])
