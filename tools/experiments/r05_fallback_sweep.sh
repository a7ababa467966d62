# the fused packet kernel's per-ray fallback: step budget sweep (OKRT_FAST_FALLBACK; 0 = off) with the packet kernel forced on every workload
mkdir -p gpurun_out/r05w
for wl in sponza dragon dragon8m dragon80k glass; do
  for fb in 0 24 48 96 192; do
    OKRT_FAST_PACKET=1 OKRT_FAST_FALLBACK=$fb python3 bench.py --workload $wl --no-cpu-baseline --no-single-call --no-repeats > gpurun_out/r05w/${wl}_fb$fb.json 2> gpurun_out/r05w/${wl}_fb$fb.err || echo "FAILED $wl $fb"
  done
done
python3 - <<'PY'
import json
print(f"{'workload':10s} {'budget':>6s} {'ms/frame':>9s} {'camera seg ms':>13s} {'bounce trace':>12s}")
for wl in ("sponza","dragon","dragon8m","dragon80k","glass"):
    for fb in (0,24,48,96,192):
        try:
            d=json.loads(open(f"gpurun_out/r05w/{wl}_fb{fb}.json").read())
        except Exception as e:
            print(wl, fb, "no line", e); continue
        ks=d["roofline"]["kernels"]
        cam=ks[0]["ms_per_frame"] if ks else 0
        tr=sum(k["ms_per_frame"] for k in ks if k["kernel"].startswith("traceKernel"))
        print(f"{wl:10s} {fb:6d} {d['ms_per_step']:9.4f} {cam:13.4f} {tr:12.4f}")
PY
