"""Per-kernel launch statistics out of a rocprofv3 results database (the .db `rocprofv3 --kernel-trace` writes).

    python tools/kstats.py gpurun_out/prof_x/x_results.db [n_rows]
"""
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    if not path.endswith(".db"):
        path = glob.glob(path + "/*.db")[0]
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]
    ks = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start) from {kd} d "
        f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    print(f"{'kernel':72s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'share':>6s}")
    for r in rows[:top]:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2] / 1e3:9.1f} {r[4] / 1e3:9.1f} {100 * r[3] / tot:5.1f}%")


if __name__ == "__main__":
    main()
