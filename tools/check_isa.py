#!/usr/bin/env python3
"""Static checks of the BUILT gfx950 code object (no recompilation: the code object is taken out of
folve_amd/csrc/build/kernels/kernels.o, or out of the library).

Why: K2's whole-call walk (kernels.hip `mac_walk_kernel<.., PIN = true, ..>`) issues its window loads by inline asm and
waits with hand-counted `s_waitcnt vmcnt(N)`.  The compiler does not know that the asm's destination register is written
asynchronously: a register copy, a spill, or one more memory instruction in the loop (a new hipcc, more register
pressure) would read stale data — intermittently, so a parity run can pass with the bug present.  The block walkers of
K1 / K3 must stay at <= 128 VGPRs or only one workgroup fits a CU.

Checks
  1. resources, from the code object's metadata notes: every mac_walk_kernel instantiation has no scratch and no
     VGPR spills; every forward_walker / inverse_walker instantiation has <= 128 VGPRs and no scratch.
  2. the walk loop of every PIN mac_walk_kernel, from the disassembly: the loop (the region closed by the kernel's last
     backward `s_branch`) is simulated twice around with the hardware's in-order VMEM counter — every instruction that
     reads or writes a register that a younger-than-waited `global_load` of the loop has as its destination is a
     hazard; `s_waitcnt vmcnt(n)` retires all but the n youngest.  Also: no scratch_ / buffer_ / flat_ access and no
     accvgpr traffic in the loop, and loads == stores == KR + D (the ring's size: one load and one store per step).

The three-FMA walk (kernels/mac_walk3.hip, `mac_walk3_kernel<KR, D, PIN, LPB, NP>`, its own object file) has the same loop
structure — one pinned load and one store per step, KR + D steps per round — and is checked the same way; its window of
(a + b) sums is written by VALU instructions only, so nothing of it is ever in flight.  Its rows are BUFFER accesses
(`buffer_load_dwordx2` / `buffer_store_dwordx2` with a scalar row offset: these two count as the walk's loads and stores,
any other buffer access is flagged), it leaves its loop once per group of steps, and the compiler may rotate the loop or
make the back edge the conditional branch: the loop is found from an `s_setprio 0` marker at its head and followed along
its hot path (`loop_region`).

usage: check_isa.py [--json] [--object PATH]      exit code 0 = clean   (default: kernels.o and mac_walk3.o)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"))


def extract_code_object(obj, workdir):
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "kernels.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(workdir, "discard.o")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                           "--input=" + fat, "--output=" + co])
    return co


def kernel_metadata(co):
    """[{name, vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size}] from the AMDGPU metadata note."""
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    kernels, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s*(- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(2), m.group(3).strip()
        if key == "agpr_count" and m.group(1):          # first key of a kernel's map (keys are sorted)
            cur = {"agpr_count": int(val)}
            kernels.append(cur)
        elif cur is not None and key in ("name", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "sgpr_count"):
            cur[key] = val if key == "name" else int(val)
    return [k for k in kernels if "name" in k]


def demangled_args(name):
    """template integers of a mangled instantiation: ..mac_walk_kernelILi33ELi7ELb1ELi6ELi1ELi1EE.. -> [33, 7, 1, 6, 1, 1]"""
    m = re.search(r"kernelI((?:L[ib]\d+E)+)E", name)
    return [int(x) for x in re.findall(r"L[ib](\d+)E", m.group(1))] if m else []


VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = []
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(ins):
    """(mnemonic, [operand strings]) of one disassembled instruction line."""
    ins = ins.split("//")[0].strip()
    parts = ins.split(None, 1)
    if not parts:
        return "", []
    if len(parts) == 1:
        return parts[0], []
    # operands are comma separated; modifiers (op_sel:[0,1]) contain commas inside brackets
    ops, depth, cur = [], 0, ""
    for ch in parts[1]:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return parts[0], ops


def function_bodies(co, want):
    """{symbol: [instruction lines]} for symbols whose name contains `want`."""
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", co], text=True)
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            sym = m.group(1)
            if re.match(r"^L\d+$", sym):
                if cur is not None:
                    out[cur].append("LABEL " + sym)
                continue
            cur = sym if want in sym else None
            if cur is not None:
                out[cur] = []
            continue
        if cur is not None and line.strip():
            out[cur].append(line.strip())
    return out


def loop_region(body):
    """The instructions of the walk loop in the order the walk executes them, or None.  mac_walk3.hip marks its loop's head
    with an `s_setprio 0`: from there the hot path is followed — a forward conditional branch inside the loop is an exit and is not taken, an s_branch
    and a conditional branch back to the head are — until it comes back to the marker.  Without a marker (kernels.hip): the loop closed by the function's last backward s_branch."""
    labels = {l.split()[1]: i for i, l in enumerate(body) if l.startswith("LABEL ")}
    marks = [i for i, l in enumerate(body) if l.startswith("s_setprio")]
    if len(marks) == 1:
        def leads_to_head(i):
            """Does straight-line code from i (following s_branch only) reach the marker?  (The compiler may rotate the loop:
            the back edge then lands some steps in front of the marker.)"""
            for _ in range(len(body)):
                if i >= len(body):
                    return False
                if i == marks[0]:
                    return True
                mn_, ops_ = split_operands(body[i])
                if mn_ == "s_endpgm":
                    return False
                # (s_cbranch_execnz: the structurizer's "always" — a walking wavefront's EXEC is never zero)
                i = labels[ops_[0]] if (mn_ in ("s_branch", "s_cbranch_execnz") and ops_ and ops_[0] in labels) else i + 1
            return False

        out, pc = [], marks[0] + 1
        while pc != marks[0] and len(out) <= len(body):
            if pc >= len(body):
                return None
            l = body[pc]
            mn, ops = split_operands(l)
            if l.startswith("LABEL "):
                pc += 1
                continue
            if mn == "s_endpgm":
                return None
            out.append(l)
            if mn in ("s_branch", "s_cbranch_execnz") and ops and ops[0] in labels:
                pc = labels[ops[0]]
            elif mn == "s_cbranch_execz":
                pc += 1
            elif mn.startswith("s_cbranch") and ops and ops[0] in labels and leads_to_head(labels[ops[0]]):
                pc = labels[ops[0]]                           # the back edge, where the compiler made it the conditional one
            else:
                pc += 1
        return out if pc == marks[0] else None
    for i in range(len(body) - 1, -1, -1):
        mn, ops = split_operands(body[i])
        if mn == "s_branch" and ops and ops[0] in labels and labels[ops[0]] < i:
            return [l for l in body[labels[ops[0]]:i + 1] if not l.startswith("LABEL ")]
    return None


def check_walk_loop(name, body):
    """Hazard simulation of the PIN walk loop.  Returns a list of problems (strings)."""
    problems = []
    args = demangled_args(name)
    if len(args) < 3 or args[2] != 1:
        return problems                      # not a PIN instantiation: compiler-scheduled loads, the compiler's own waits
    kr, d = args[0], args[1]
    loop = loop_region(body)
    if loop is None:
        return ["%s: no backward s_branch found — the walk loop's shape changed, update tools/check_isa.py" % name]
    loads = stores = 0
    inflight = []                            # oldest first: the destination registers of a load, or None for a store
    for rnd in range(2):
        for ins in loop:
            mn, ops = split_operands(ins)
            if not mn:
                continue
            # (the three-FMA walk addresses its rows as buffer accesses — a fixed resource and a scalar row offset — so
            # buffer_load_dwordx2 / buffer_store_dwordx2 ARE its loads and stores; any other buffer access is unexpected)
            if mn in ("buffer_load_dwordx2", "buffer_store_dwordx2"):
                mn = mn.replace("buffer_", "global_")
            if mn.startswith(("scratch_", "buffer_", "flat_")) or "accvgpr" in mn:
                if rnd == 0:
                    problems.append("%s: `%s` inside the walk loop" % (name, ins.split("//")[0].strip()))
                continue
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ins)
                if m:
                    n = int(m.group(1))
                    while len(inflight) > n:
                        inflight.pop(0)
                continue
            if mn.startswith("global_load"):
                dst = vregs(ops[0])
                srcs = [r for o in ops[1:] for r in vregs(o)]
                pending = {r for e in inflight if e for r in e}
                for r in srcs + dst:
                    if r in pending:
                        problems.append("%s: `%s` touches v%d while a load into it is in flight" % (name, ins.split("//")[0].strip(), r))
                inflight.append(tuple(dst))
                loads += rnd == 0
                if not mn.endswith("dwordx2") and rnd == 0:
                    problems.append("%s: unexpected `%s` in the walk loop" % (name, mn))
                continue
            if mn.startswith("global_store") or mn.startswith("global_atomic"):
                pending = {r for e in inflight if e for r in e}
                for r in [r for o in ops for r in vregs(o)]:
                    if r in pending:
                        problems.append("%s: `%s` reads v%d while a load into it is in flight" % (name, ins.split("//")[0].strip(), r))
                inflight.append(None)
                stores += rnd == 0
                continue
            touched = [r for o in ops for r in vregs(o)]
            if touched:
                pending = {r for e in inflight if e for r in e}
                for r in touched:
                    if r in pending:
                        problems.append("%s: `%s` uses v%d before the load into it was waited for" % (name, ins.split("//")[0].strip(), r))
    if loads != kr + d or stores != kr + d:
        problems.append("%s: %d loads and %d stores in the walk loop, %d each expected (KR + D)" % (name, loads, stores, kr + d))
    problems += vcc_writers_in_loop(name, loop)
    return sorted(set(problems))


def vcc_writers_in_loop(name, loop):
    """mac_walk3.hip's hand-down (`v_cndmask_b32_dpp .., vcc`, several lanes per path) reads the lane sets' heads from VCC, which
    the loop's head sets ONCE per round (`s_mov_b64 vcc` right behind the marker).  Between that and the round's last
    hand-down nothing may write VCC: no other instruction naming vcc, no VOP2 / VOPC form that writes it implicitly."""
    users = [i for i, ins in enumerate(loop) if split_operands(ins)[0] == "v_cndmask_b32_dpp" and re.search(r"\bvcc\b", ins)]
    if not users:
        return []
    problems = []
    mn0, ops0 = split_operands(loop[0])
    if not (mn0 == "s_mov_b64" and ops0 and ops0[0] == "vcc"):
        problems.append("%s: the walk loop's hand-downs read vcc but the loop's head does not set it (`%s`)" % (name, loop[0].split("//")[0].strip()))
    for ins in loop[1:users[-1]]:
        mn, ops = split_operands(ins)
        if mn == "v_cndmask_b32_dpp":
            continue
        implicit = (mn.startswith("v_cmp") and mn.endswith("_e32")) or (mn.endswith("_e32") and ("_co_" in mn or mn.startswith(("v_addc", "v_subb", "v_div_fmas"))))
        if implicit or re.search(r"\bvcc(_lo|_hi)?\b", ins.split("//")[0]):
            problems.append("%s: `%s` touches vcc between the loop's head and its last hand-down" % (name, ins.split("//")[0].strip()))
    return problems


WALK_NAMES = ("mac_walk_kernel", "mac_walk3_kernel", "mac_walk3_nt_kernel")   # kernels.hip (four FMAs per complex MAC), mac_walk3.hip (three; _nt: rows with the non-temporal hint)


def default_objects():
    d = os.path.join(ROOT, "folve_amd", "csrc", "build", "kernels")
    return [os.path.join(d, "kernels.o"), os.path.join(d, "mac_walk3.o")]


def run(obj=None):
    """obj: one object file, a list of them, or None = every object that holds walk kernels."""
    if not tools_present():
        return {"skipped": "llvm tools not found under " + LLVM}
    objs = default_objects() if obj is None else ([obj] if isinstance(obj, str) else list(obj))
    for o in objs:
        if not os.path.exists(o):
            return {"skipped": o + " not built"}
    problems, report = [], {"object": [os.path.relpath(o, ROOT) for o in objs], "kernels": {}}
    n_walk = n_walker = checked = 0
    for o in objs:
        with tempfile.TemporaryDirectory() as tmp:
            co = extract_code_object(o, tmp)
            meta = kernel_metadata(co)
            for k in meta:
                name = k["name"]
                if any(w in name for w in WALK_NAMES):
                    n_walk += 1
                    if k.get("private_segment_fixed_size", 0) != 0 or k.get("vgpr_spill_count", 0) != 0:
                        problems.append("%s: scratch %d bytes, %d VGPR spills" % (name, k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0)))
                    report["kernels"][name] = {"vgprs": k.get("vgpr_count"), "args": demangled_args(name)}
                elif "forward_walker_kernel" in name or "inverse_walker_kernel" in name:
                    n_walker += 1
                    if k.get("vgpr_count", 0) > 128 or k.get("private_segment_fixed_size", 0) != 0:
                        problems.append("%s: %d VGPRs (budget 128: two workgroups per CU), scratch %d bytes" %
                                        (name, k.get("vgpr_count", 0), k.get("private_segment_fixed_size", 0)))
                    report["kernels"][name] = {"vgprs": k.get("vgpr_count")}
            for want in WALK_NAMES:
                for name, body in function_bodies(co, want).items():
                    if name.endswith(".kd"):
                        continue
                    problems.extend(check_walk_loop(name, body))
                    a = demangled_args(name)
                    if len(a) >= 3 and a[2] == 1:
                        checked += 1
    if n_walk == 0 or n_walker == 0:
        problems.append("no mac_walk_kernel / walker instantiations found in the code objects (%d / %d)" % (n_walk, n_walker))
    report["walk_kernels"] = n_walk
    report["walker_kernels"] = n_walker
    report["walk_loops_simulated"] = checked
    report["pinned"] = checked > 0
    report["problems"] = problems
    report["ok"] = not problems
    return report


def main():
    obj = None
    if "--object" in sys.argv:
        obj = sys.argv[sys.argv.index("--object") + 1]
    rep = run(obj)
    if "--json" in sys.argv:
        print(json.dumps(rep))
    else:
        if "skipped" in rep:
            print("check_isa: skipped (%s)" % rep["skipped"])
            return 0
        for p in rep["problems"]:
            print("check_isa: " + p)
        print("check_isa: %d mac_walk kernels (%d walk loops simulated), %d walkers: %s" %
              (rep["walk_kernels"], rep["walk_loops_simulated"], rep["walker_kernels"], "clean" if rep["ok"] else "PROBLEMS"))
    return 0 if rep.get("ok", True) else 1


if __name__ == "__main__":
    sys.exit(main())
