"""Issue-order trace of the memory operations of compiled kernels (development aid, no GPU needed).

Compiles rootba_amd/csrc/solver.hip to gfx950 assembly and prints, per requested kernel, the sequence of
  Ln  global / buffer load of n dwords        S  global store        A  global atomic        d  LDS operation
  Wn  s_waitcnt vmcnt(n)                      w  s_waitcnt lgkmcnt   B  branch               BAR  s_barrier
  <k> k other instructions in between         |label|  basic-block labels
A kernel that is a chain of memory round trips shows up as `L W0 ... L W0 ...` (a load, a wait for everything, the next
load): loads inside conditionals, loads behind may-alias stores. DESIGN.md 3a ("Memory round trips, not bytes") lists what
this found in round 3.

usage: python scripts/isa_trace.py 'k_cam_pass_mfma<float, 0>' 'k_bs_tile<float>' 'k_pcgs_*' ...
       (names as c++filt prints them without the argument list; a trailing * matches a prefix)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly():
    out = os.path.join(tempfile.gettempdir(), "rootba_isa_trace.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S",
                           "--cuda-device-only", "solver.hip", "-o", out], cwd=os.path.join(ROOT, "rootba_amd", "csrc"),
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def trace(body):
    out, other = [], 0

    def flush():
        nonlocal other
        if other:
            out.append(str(other))
            other = 0

    for line in body.splitlines():
        p = line.split()
        if not line.startswith("\t") or not p or p[0].startswith((".", ";")):
            if re.match(r"\.LBB", line):
                flush()
                out.append("|" + line.split(":")[0].strip() + "|")
            continue
        op = p[0]
        if op.startswith(("global_load", "buffer_load")):
            flush()
            m = re.search(r"dwordx(\d)", op)
            out.append("L" + (m.group(1) if m else "1"))
        elif op.startswith("global_store"):
            flush()
            out.append("S")
        elif op.startswith("global_atomic"):
            flush()
            out.append("A")
        elif op.startswith("ds_"):
            flush()
            out.append("d")
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:
                flush()
                out.append("W" + m.group(1))
            elif "lgkmcnt" in line:
                flush()
                out.append("w")
        elif op.startswith("s_cbranch") or op == "s_branch":
            flush()
            out.append("B")
        elif op == "s_barrier":
            flush()
            out.append("BAR")
        else:
            other += 1
    flush()
    return " ".join(out)


if __name__ == "__main__":
    want = sys.argv[1:]
    if not want:
        sys.exit(__doc__)
    text = assembly()
    names = re.findall(r"\n(_ZN3rba[^\n:]+):[^\n]*\n", text)
    demangled = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for name, dem in zip(names, demangled):
        short = dem.split("(")[0].replace("void rba::", "").replace("rba::", "")
        if not any(w == short or (w.endswith("*") and short.startswith(w[:-1])) for w in want):
            continue
        i = text.index("\n" + name + ":")
        j = text.index(".Lfunc_end", i)  # (a kernel may hold several s_endpgm: early exits)
        print("==", short)
        print(trace(text[i:j]))
