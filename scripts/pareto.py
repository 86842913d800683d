"""Time-to-loss table for the knobs of the fused hogwild kernel at config C: for every variant, bench.py's examples/s,
hold-out log-loss curve and seconds to the target, over `reps` runs.  The shipped default must be Pareto-best on (examples/s, loss at equal
examples), or the table says which variant should be.
usage: python scripts/pareto.py [reps=3] [steps=20] [warmup=5]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = sys.argv[2] if len(sys.argv) > 2 else "20"
warm = sys.argv[3] if len(sys.argv) > 3 else "5"
variants = [
    ("default", [], {}),
    ("in_flight_512", ["--max-in-flight", "512"], {}),
    ("in_flight_384", ["--max-in-flight", "384"], {}),
    ("in_flight_256", ["--max-in-flight", "256"], {}),
    ("round1_path", ["--whole-lines", "0"], {}),
    ("round1_path_512", ["--whole-lines", "0", "--max-in-flight", "512"], {}),
    # build variants (scripts/build_variant.sh maxrNwin -DFW_MAXR_WIN=N): rows per wave kept from the gather (the shipped build: 2)
    ("t1024m14", ["--threads", "1024", "--wgs-per-cu", "1"], {"FWGPU_LIBRARY": os.path.join(ROOT, "build", "variants", "libfwgpu_t1024m14.so")}),
    ("t640m10", ["--threads", "640", "--wgs-per-cu", "2"], {"FWGPU_LIBRARY": os.path.join(ROOT, "build", "variants", "libfwgpu_t640m10.so")}),
] + [(f"w4m{m}", [], {"FWGPU_LIBRARY": os.path.join(ROOT, "build", "variants", f"libfwgpu_w4m{m}.so")}) for m in (12, 14, 16, 20, 24)] + [(v, [], {"FWGPU_LIBRARY": os.path.join(ROOT, "build", "variants", f"libfwgpu_{v}.so")}) for v in ("w4m12uo2", "w4m12uo2ua4", "w4m12ug8", "w4m16uo2")] + [(f"maxr{m}win", [], {"FWGPU_LIBRARY": os.path.join(ROOT, "build", "variants", f"libfwgpu_maxr{m}win.so")}) for m in (0, 1, 2, 3, 4, 6, 8, 10, 12, 16)] + [
    ("whole_lines_2", ["--whole-lines", "2"], {}),
    ("placement_off", [], {"FWGPU_PLACEMENT": "0"}),
    ("no_chain", [], {"FWGPU_BENCH_NO_CHAIN": "1"}),
    ("hot_lr_plain", [], {"FWGPU_HOT_LR_EVERY": "0"}),
    ("hot_lr_every32", [], {"FWGPU_HOT_LR_EVERY": "32"}),
]
if os.environ.get("PARETO_ONLY"):
    keep = set(os.environ["PARETO_ONLY"].split(","))
    variants = [v for v in variants if v[0] in keep]
print(f"{'variant':18s} {'ex/s (M)':>24s} {'final hold-out':>26s} {'s to target':>22s}  curve of the last run")
for name, flags, env in variants:
    vals, lls, secs, curve, place = [], [], [], None, []
    for _ in range(reps):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", warm, "--no-traffic", "--no-cpu-baseline", "--target-logloss", os.environ.get("PARETO_TARGET", "0.655")] + flags,
                             env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        vals.append(d["value"] / 1e6)
        lls.append(d["final_logloss"])
        secs.append(d["seconds_to_logloss"]["seconds"])
        curve = {k: round(v, 4) for k, v in d["logloss_after_examples"].items()}
        place.append(d["table_placement"]["candidates_tried"])
    print(f"{name:18s} {' '.join(f'{v:.3f}' for v in vals):>24s} {' '.join(f'{v:.4f}' for v in lls):>26s} "
          f"{' '.join('-' if v is None else f'{v:.3f}' for v in secs):>22s}  placement tries {place}  {curve}", flush=True)
