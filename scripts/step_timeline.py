"""Print the per-kernel timeline of the last training step from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof*/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(path)) if "suave" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(rows) if "k_expand" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void suave::", "").replace("suave::", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    print(f"{name:36s} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']:>4s} "
          f"start={(int(r['Start_Timestamp']) - t0) / 1e3:8.1f}us dur={dur:7.1f}us vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']} lds={r['LDS_Block_Size']}")
print("sum of kernel durations: %.1f us" % tot)
