mkdir -p gpurun_out/r05e
for wl in sponza dragon dragon8m dragon80k glass; do
  for pk in 1 0; do
    OKRT_FAST_PACKET=$pk python3 bench.py --workload $wl --no-cpu-baseline --no-single-call --no-repeats > gpurun_out/r05e/${wl}_packet$pk.json 2> gpurun_out/r05e/${wl}_packet$pk.err || echo "FAILED $wl $pk"
  done
done
python3 - <<'PY'
import json
print("# camera segment: packet kernel (default, fused) vs per-ray (OKRT_FAST_PACKET=0: generate + traceKernel + shadeKernel<FIRST>); bench.py --workload W --no-cpu-baseline, 128 steps, 64-frame sets")
print(f"{'workload':10s} {'mode':8s} {'ms/frame':>9s} {'camera seg ms':>13s} {'bounce trace':>12s} {'later shade':>11s} {'tri/cam tile':>12s} {'tri/bounce ray':>14s}")
for wl in ("sponza","dragon","dragon8m","dragon80k","glass"):
    for pk in (1,0):
        try:
            d=json.loads(open(f"gpurun_out/r05e/{wl}_packet{pk}.json").read())
        except Exception as e:
            print(wl, pk, "no line", e); continue
        r=d["roofline"]; ks=r["kernels"]
        cam=ks[0]["ms_per_frame"] if ks else 0
        tr=sum(k["ms_per_frame"] for k in ks[1:] if k["kernel"].startswith("traceKernel"))  # (ks[0] is the camera segment: "traceKernelCamera + ..." when it runs per ray)
        sh=sum(k["ms_per_frame"] for k in ks if k["kernel"].startswith(("missKernel","shadeKernel")))
        print(f"{wl:10s} {'packet' if pk else 'per-ray':8s} {d['ms_per_step']:9.4f} {cam:13.4f} {tr:12.4f} {sh:11.4f} {r['tri48_tests_per_camera_tile']:12.2f} {r['tri48_tests_per_bounce_ray']:14.2f}")
PY
