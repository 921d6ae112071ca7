"""Print the per-kernel totals of a rocprofv3 rocpd database (the *_results.db of `rocprofv3 --kernel-trace --stats`),
own kernels (libsnws) and foreign ones (torch / vendor libraries) separately; optionally write them as CSV."""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
    rows = [dict(zip(cols, r)) for r in cur.execute("select * from top_kernels")]
    tot = sum(r["total_duration"] for r in rows)
    own = sum(r["total_duration"] for r in rows if "anonymous namespace" in r["name"] or "_GLOBAL__N_" in r["name"])
    print("kernels %d, total %.3f ms, libsnws share %.1f%%" % (len(rows), tot / 1e3, 100.0 * own / tot))   # the view reports microseconds
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            for r in rows:
                w.writerow([r["name"], r["total_calls"], int(round(r["total_duration"] * 1e3)), r["average"] * 1e3, r["percentage"]])
    for r in rows:
        mine = "anonymous namespace" in r["name"] or "_GLOBAL__N_" in r["name"]
        if r["percentage"] >= 0.3 or (not mine and r["percentage"] >= 0.05):
            print("%s %-110s calls %5d  %9.3f ms  %5.2f%%" % ("   " if mine else "EXT", r["name"][:110], r["total_calls"],
                                                               r["total_duration"] / 1e3, r["percentage"]))


if __name__ == "__main__":
    main()
