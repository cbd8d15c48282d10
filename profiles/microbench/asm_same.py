"""Do two builds of csrc/fpx_api.hip hold the same machine code, function by function?

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --cuda-device-only -S -o before.s frankenpaxos_amd/csrc/fpx_api.hip
    ... edit ...
    hipcc (the same) -o after.s frankenpaxos_amd/csrc/fpx_api.hip
    python profiles/microbench/asm_same.py before.s after.s [substring of the mangled names to look at]

Compares the instruction text of every function both files define (comments, directives and the numbering of local labels
dropped).  Used in round 5 to move the vote kernel's body into fpx_phase2_body.inc (a second kernel includes it) and the
division by multiplication into fpx_fastdiv.hpp WITHOUT touching what had been measured: all 130 k_phase2 instantiations
resp. all 186 device functions came out identical; wrapping the body in a __device__ function instead had changed every one
of the 130 (profiles/r05_cfg5.md)."""
import re
import sys


def functions(path):
    out, name, buf = {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, buf = m.group(1), []
            out[name] = buf
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        text = line.split(";")[0].rstrip()
        bare = text.strip()
        if not bare or (bare.startswith(".") and not bare.startswith(".LBB")):
            continue
        buf.append(re.sub(r"\.LBB\d+_", ".LBB_", text))
    return out


def main():
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    pattern = sys.argv[3] if len(sys.argv) > 3 else ""
    same, different = 0, []
    for name, body in a.items():
        if pattern not in name:
            continue
        if name not in b:
            print("only in", sys.argv[1], ":", name)
        elif body == b[name]:
            same += 1
        else:
            different.append((name, len(body), len(b[name])))
    for name in b:
        if pattern in name and name not in a:
            print("only in", sys.argv[2], ":", name)
    print("same", same, "different", len(different))
    for d in different[:50]:
        print("  %s: %d -> %d instructions" % d)
    return 1 if different else 0


if __name__ == "__main__":
    sys.exit(main())
