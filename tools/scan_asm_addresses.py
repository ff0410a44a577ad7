"""Compiles every device translation unit of csrc/ to ISA text (hipcc -S --cuda-device-only, the build's own flags) and scans it for
a mis-scheduling seen once in round 6: a 64-bit address built by s_add_u32 / s_addc_u32 whose add-with-carry the scheduler placed
BEHIND the inline-asm memory instruction that reads the pair (the instruction then used a half-made address; an empty
`asm volatile("" : "+s"(addr))` in front pins the pair).  A suspect = an s_addc_u32 into a pair's high half right behind a memory
instruction that used the pair, with no s_add_u32 of the low half in between.   python3 tools/scan_asm_addresses.py [outdir]"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "signaloperators.jl_amd", "csrc")
sys.path.insert(0, C)
import build  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sigops_isa"
os.makedirs(out, exist_ok=True)


def compile_unit(s):
    unit = None
    name = s.replace("@", "_u").replace(".hip", "")
    if "@" in s:
        s, unit = s.split("@")
    cmd = ["/opt/rocm/bin/hipcc", "-mllvm", "-disable-machine-licm", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
           "-Wno-unused-function", "-x", "hip", "--cuda-device-only", "-S", os.path.join(C, s), "-o", os.path.join(out, name + ".s")]
    if unit is not None and s == "k_resample.hip":
        cmd[1:1] = ["-DSO_RP_UNIT=" + unit]
    elif unit is not None:
        m = re.fullmatch(r"(\d+)(?:([df])(\d+))?", unit)
        cmd[1:1] = ["-DSO_RSOS_ONLY_KS=" + m.group(1)] + (["-DSO_RSOS_ONLY_F32=%d" % (m.group(2) == "f"), "-DSO_RSOS_ONLY_NW=" + m.group(3)] if m.group(2) else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(out, name + ".s")


def scan(path):
    lines = open(path).read().splitlines()
    bad, n = [], 0
    for i, l in enumerate(lines):
        m = re.search(r"\b(global_load\w*|global_store\w*|global_atomic\w*|s_load\w*|buffer_load\w*)\s.*\bs\[(\d+):(\d+)\]", l)
        if not m or int(m.group(3)) != int(m.group(2)) + 1:
            continue
        n += 1
        lo, hi = int(m.group(2)), int(m.group(3))
        for j in range(i + 1, min(i + 8, len(lines))):
            t = lines[j].strip()
            if t.startswith(";;#ASM") or t == "":
                continue
            if re.match(r"s_add_u32 s%d," % lo, t):
                break
            if re.match(r"s_addc_u32 s%d," % hi, t):
                bad.append((i + 1, l.strip(), t))
                break
            if not t.startswith("s_"):
                break
    return n, bad


units = [s for s in build.SRCS if s.split("@")[0].endswith(".hip")]
with ThreadPoolExecutor(max_workers=8) as pool:
    paths = list(pool.map(compile_unit, units))
total, suspects = 0, 0
for p in paths:
    n, bad = scan(p)
    total += n
    suspects += len(bad)
    for b in bad:
        print("SUSPECT", os.path.basename(p), *b)
print({"units": len(paths), "memory_instructions_with_a_scalar_address_pair": total, "suspects": suspects})
