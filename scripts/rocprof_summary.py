"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) results into small text summaries for profiles/.
usage: rocprof_summary.py <trace.db> [<pmc.db> ...] > profiles/rNN_xxx.txt"""
import sqlite3, sys
for path in sys.argv[1:]:
    con = sqlite3.connect(path); cur = con.cursor()
    print(f"## {path}")
    try:
        rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        if rows:
            print("# kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total_us, avg_us, pct")
            for r in rows: print(f"{r[0]}, {r[1]}, {r[2]:.1f}, {r[3]:.1f}, {r[4]:.2f}")
    except sqlite3.Error as e:
        print("no top_kernels:", e)
    try:
        rows = list(cur.execute("select kernel_name, counter_name, value, duration, grid_size, workgroup_size, lds_block_size, vgpr_count, sgpr_count from counters_collection where kernel_name like '%fw_example_kernel%'"))
        if rows:
            print("# PMC per dispatch (FETCH_SIZE / WRITE_SIZE are in KB): kernel, counter, value, duration_us, grid, wg, lds, vgpr, sgpr")
            for r in rows: print(f"{r[0]}, {r[1]}, {r[2]:.1f}, {r[3]/1e3:.1f}, {r[4]}, {r[5]}, {r[6]}, {r[7]}, {r[8]}")
    except sqlite3.Error as e:
        print("no counters_collection:", e)
    print()
