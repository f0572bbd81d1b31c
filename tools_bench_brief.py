import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "ms/step", d["ms_per_step"], "Mray/s", d["value"], "| excl:", {k: v["ms_per_frame"] for k, v in r["kernels"].items()}, "1-stream frame", r["frame_ms_one_stream"],
              "| gather:", d["gather_per_frame"], "| l2", r["kernels"]["k_gather"]["l2_frac"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-400:])
