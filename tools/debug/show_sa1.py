import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"])
for r in d["roofline"]["binding"]["by_lost_time"]:
    if "131072" in r["launch"]: print("   %-62s ms %.3f gbs %.0f"%(r["launch"],r["ms_per_step"],r["algorithmic_gbs"]))
