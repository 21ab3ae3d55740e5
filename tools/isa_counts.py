"""Instruction-kind counts per k_trace instantiation of a device assembly listing
(hipcc -S --cuda-device-only ... -o trace.s).  usage: python tools/isa_counts.py trace.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
for m in re.finditer(r"^(_ZN4rtgs\S*?):\s*; @", txt, flags=re.M):
    n = m.group(1)
    if flt not in n:
        continue
    end = txt.index("s_endpgm", m.end())
    end2 = txt.find(".Lfunc_end", end)
    f = txt[m.end():end2]
    tp = re.search(r"k_traceI(.*?)EEv", n)
    print((tp.group(1) if tp else n)[:60].ljust(62), {k: f.count(k) for k in ("s_load_dwordx16", "s_load_dwordx8", "global_load_dwordx4", "global_load_dword ", "scratch_", "v_readlane", "v_writelane", "s_cbranch")}, "lines", f.count("\n"))
