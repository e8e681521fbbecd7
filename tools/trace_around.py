"""Which kernels surround the slow runtime blit kernels (__amd_rocclr_copyBuffer > 50 us) in a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py ... ; python tools/trace_around.py DIR
(Round 2: the 620 us copies in a bench trace are the device-to-host stencil checks of pgx_create, i.e. setup, not the timed solve.)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "copyBuffer" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000]
print(len(idx), "slow copyBuffer launches")
for i in idx[3:6]:
    for r in rows[max(0, i - 4): i + 4]:
        print("   ", r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
    print("---")
