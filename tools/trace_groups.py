"""Per (kernel, grid) durations of a rocprofv3 --kernel-trace csv.  usage: python tools/trace_groups.py <kernel_trace.csv>"""
import collections, csv, sys
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = [x for x in ("k_trace", "k_project", "k_bin", "k_untile", "k_upload", "k_peel") if x in r["Kernel_Name"]]
    if k:
        g[(k[0], r["Grid_Size_X"], r["Grid_Size_Y"], "vgpr " + r["VGPR_Count"], "scratch " + r["Scratch_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(g.items()):
    v2 = sorted(v)
    print(*k, "calls", len(v), "avg %.1f us  med %.1f  min %.1f" % (sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3))
