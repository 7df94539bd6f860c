import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][0]
d=json.loads(l)
print("value",d["value"],"ms",d["ms_per_step"],"median",d.get("ms_per_step_median_rank0"))
r=d["roofline"]; print("frac",r["frac"],"prof",r.get("frac_from_profile"),r["kernel"])
print("f32 instr: ms/step", r.get("f32_instr_ms_per_step"), "kernel frac", r.get("f32_instr_kernel_frac"), "from profile",
      r.get("f32_instr_frac_from_profile"), "pmc passes", r.get("f32_instr_frac_from_profile_pmc_passes"), "rel err", r.get("f32_instr_rel_err"))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
for k,v in d.get("configs",{}).items():
    ro=v.get("roofline") or {}
    print(k, "ms/step", v.get("ms_per_step"), "value", v.get("value"), "bound", ro.get("bound"), "frac", ro.get("frac"), "mfma alg/issued", ro.get("mfma_algorithmic_frac"), ro.get("mfma_issued_frac"), ro.get("kernel"), "cpu", (v.get("cpu_baseline") or {}).get("value"), v.get("error"), (v.get("result_on_device") or {}).get("ms_per_step"))
for k in ("small_tiles","crystallinity","live_feed","host_streamed","delivery_anchor_n1"):
    print(k, json.dumps(d.get(k))[:600])
