"""Register pressure of ONE instantiation of the example kernels, in about half a minute instead of the four of a full build (no GPU needed):
a scratch copy of kernels.hip whose launchers reference no kernel, plus one reference to the instantiation asked for, compiled for the device only.
usage: python3 scripts/hot_probe.py ['fw_example_kernel_r<300, true, 20, true, 1, 3>'] [-DFLAG ...]      (default: the shipped config-C instantiation)
       PHASE=1 compiles the second translation unit (-DFW_PHASE_TU) instead.
Prints the resource-usage remarks and leaves the ISA in /tmp/hot_probe.s."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "fwumious_wabbit_amd", "csrc")
obj = os.path.join(ROOT, "fwumious_wabbit_amd", "lib", "obj")
os.makedirs(obj, exist_ok=True)
args = [a for a in sys.argv[1:] if not a.startswith("-")]
flags = [a for a in sys.argv[1:] if a.startswith("-")]
kern = args[0] if args else "fw_example_kernel_r<300, true, 20, true, 1, 3>"
s = open(os.path.join(src, "kernels.hip")).read()
s, n = re.subn(r"launch_persistent\(fw_example_kernel(?:_r)?<[^;]*?>, p,", "launch_none(p,", s)
s = s.replace("template <typename K>\nstatic hipError_t launch_persistent(", "static hipError_t launch_none(const KernelParams &, uint32_t, uint32_t, size_t, hipStream_t) { return hipErrorInvalidValue; }\n"
              "template <typename K>\nstatic hipError_t launch_persistent(", 1)
assert n >= 19 and "launch_none(const" in s, n
s += f"\nnamespace fwgpu {{ hipError_t probe_launch(const KernelParams &p, hipStream_t st) {{ return launch_persistent({kern}, p, 1, 512, 0, st); }} }}\n"
path = os.path.join(obj, f"kernels_probe_{os.getpid()}.hip")  # (per process: the register-budget tests may run in parallel)
asm = f"/tmp/hot_probe_{os.getpid()}.s"
open(path, "w").write(s)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{src}", "-mllvm", "-pragma-unroll-threshold=131072",
       "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + (["-DFW_PHASE_TU"] if os.environ.get("PHASE") else []) + flags + ["-S", path, "-o", asm]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
os.remove(path)
if os.path.exists(asm):
    os.replace(asm, "/tmp/hot_probe.s")
keep = False
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
    if m:
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            keep = "fw_example_kernel" in t
            if keep:
                print(subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip())
        elif keep and any(k in t for k in ("SGPRs", "VGPRs", "Scratch", "Occupancy")):
            print("   ", t)
    elif "error" in line:
        print(line)
