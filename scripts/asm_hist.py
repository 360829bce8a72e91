"""Static instruction histogram of a kernel in the gfx950 asm (hipcc -S --cuda-device-only)."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "k_linearizeILi0ELi1"
asm = "/tmp/dcreg_ctx.s"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "-o", asm, "--cuda-device-only",
                       os.path.join(ROOT, "dcreg_amd/csrc/device/context.hip")], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5dcreg.*%s.*:" % pat, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
print(lines[start][:70], "static instructions:", len(ins))
for k, v in c.most_common(40):
    print("  %-28s %d" % (k, v))
print("f64:", sum(v for k, v in c.items() if "f64" in k), " div_scale:", c["v_div_scale_f64"], " rcp:", c["v_rcp_f64_e32"],
      " rsq:", c.get("v_rsq_f64_e32", 0), " sqrt:", c.get("v_sqrt_f64_e32", 0), " cndmask:", c["v_cndmask_b32_e32"] + c["v_cndmask_b32_e64"])
