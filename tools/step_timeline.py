"""Print one step's kernel timeline from a rocprofv3 kernel trace: tools/step_timeline.py <trace.csv> [marker]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_multi"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 100000 * (marker == "adam_multi")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{r['Queue_Id']} grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>5} {n}")
