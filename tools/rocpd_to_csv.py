#!/usr/bin/env python3
"""rocprofv3's default output is a rocpd SQLite database; the summarizers under profiles/ read the kernel-trace CSV
layout.  This writes the `kernels` view of a *_results.db as that CSV (same column names), so a trace taken without
`--output-format csv` is still usable.

    python3 tools/rocpd_to_csv.py gpurun_out/x/ns_results.db gpurun_out/x/ns_kernel_trace.csv
"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, lds_size, "
                   "vgpr_count, accum_vgpr_count, sgpr_count, stream_id, queue_id from kernels order by start")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z",
                "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z", "LDS_Block_Size", "VGPR_Count",
                "Accum_VGPR_Count", "SGPR_Count", "Stream_Id", "Queue_Id"])
    n = 0
    for r in rows:
        w.writerow(r)
        n += 1
print(f"{n} kernel dispatches -> {out}")
