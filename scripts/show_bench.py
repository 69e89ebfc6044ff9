import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["mode_T"]["value"], d["shard_125MB"]["value"])
for k,v in d["adversarial"].items(): print(k, v["value"], v["roundtrip"])
for k,v in d["other_configs"].items(): print(k, v["value"], v["roundtrip"])
