import json, sys
for f in sys.argv[1:]:
    l=[x for x in open(f) if x.startswith('{"metric')][-1]
    e=json.loads(l); op=e["one_pass"]
    print(f, "one pass", op["reads_per_s"], "busy", op.get("gpu_busy_fraction_estimate"), "resident", e["resident"]["reads_per_s"], "batches only", e["resident"]["reads_per_s_batches_only"], "gpu_kernels", e["resident"]["stage_sums_s"]["gpu_kernels"], "streamed", e["streamed"]["reads_per_s"])
