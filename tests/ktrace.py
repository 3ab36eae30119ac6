"""Calls of the kernels whose name contains PATTERN in a rocprofv3 kernel_trace.csv: python tests/ktrace.py FILE PATTERN [max]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[:top]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mmt::", "")
    print("%-40s %10.1f us  grid %s  lds %s  vgpr %s" % (name[-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                        r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
