import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ib=d["index_build"]
print("index_build wall %.2f device %.2f library %.2f" % (ib["seconds"], ib["device_seconds"], ib["library_seconds"]))
for ph in ib["phases"]: print("  ", ph["name"], round(ph["ms"],1))
