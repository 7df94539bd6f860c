import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][0]
d=json.loads(l)
print("value",d["value"],"ms",d["ms_per_step"],"median",d.get("ms_per_step_median_rank0"))
r=d["roofline"]; print("frac",r["frac"],"prof",r.get("frac_from_profile"),r["kernel"])
f=d["roofline"].get("f32_instruction") or {}
print("f32", f.get("ms_per_step"), f.get("kernel_frac"), f.get("frac_from_profile"))
for k,v in d.get("configs",{}).items():
    ro=v.get("roofline") or {}
    print(k, "ms/step", v.get("ms_per_step"), "value", v.get("value"), "frac", ro.get("frac"), "hbm_frac", ro.get("hbm_frac"), ro.get("kernel"), v.get("error"), (v.get("result_on_device") or {}).get("ms_per_step"))
for k in ("small_tiles","crystallinity","live_feed","host_streamed"):
    print(k, json.dumps(d.get(k))[:600])
