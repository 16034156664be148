"""CPU: static checks of the built gfx950 code object (tools/check_isa.py).

K2's whole-call walk (`mac_walk_kernel<.., PIN = true, ..>`, folve_amd/csrc/kernels/kernels.hip; the three-FMA form
`mac_walk3_kernel`, kernels/mac_walk3.hip, with buffer-addressed rows) issues its window loads by inline asm and waits with
hand-counted `s_waitcnt vmcnt(N)`; it is only correct while no window register is spilled or
copied and the loop holds exactly the memory instructions the counts assume.  A parity run can pass with that broken (the
stale read depends on timing), so the condition is checked on the ISA itself: resource notes of every instantiation and a
simulation of every walk loop against the hardware's in-order memory counter.  hipcc cross-compiles in the build
container, so this runs in the CPU suite on the very object `build()` produced."""
import os
import re
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa  # noqa: E402

OBJ = os.path.join(ROOT, "folve_amd", "csrc", "build", "kernels", "kernels.o")
OBJ3 = os.path.join(ROOT, "folve_amd", "csrc", "build", "kernels", "mac_walk3.o")
pytestmark = pytest.mark.skipif(not (check_isa.tools_present() and os.path.exists(OBJ) and os.path.exists(OBJ3)),
                                reason="needs the ROCm llvm tools and the built kernels.o / mac_walk3.o")


def test_walk_kernels_and_walkers_are_clean():
    rep = check_isa.run()
    assert rep["ok"], "\n".join(rep["problems"][:40])
    assert rep["walk_kernels"] >= 30 and rep["walk_loops_simulated"] == rep["walk_kernels"]   # every instantiation of both forms is PIN
    assert sum("mac_walk3_kernel" in n for n in rep["kernels"]) >= 15
    assert rep["walker_kernels"] >= 4
    for name, k in rep["kernels"].items():
        if "walker_kernel" in name:
            assert k["vgprs"] <= 128, (name, k)                   # two workgroups per CU


def test_the_checker_sees_what_it_is_there_for():
    """Mutations of a real walk loop: every wait one load too lenient, a copy of a window register right behind its load,
    one more load in the loop, a scratch access — each must be reported (and the unmutated loop must not)."""
    bodies = {}
    for obj, want in ((OBJ, "mac_walk_kernel"), (OBJ3, "mac_walk3_kernel")):
        with tempfile.TemporaryDirectory() as tmp:
            co = check_isa.extract_code_object(obj, tmp)
            bodies.update({n: b for n, b in check_isa.function_bodies(co, want).items() if not n.endswith(".kd")})
    assert any("mac_walk3_kernel" in n for n in bodies) and any("mac_walk_kernel" in n for n in bodies)
    for name, body in bodies.items():
        assert check_isa.check_walk_loop(name, body) == []
        loop = set(check_isa.loop_region(body))

        def weaken(l):
            m = re.search(r"vmcnt\((\d+)\)", l)
            return l.replace(m.group(0), "vmcnt(%d)" % (int(m.group(1)) + 1)) if (m and l in loop) else l

        assert check_isa.check_walk_loop(name, [weaken(l) for l in body])
        idx = [i for i, l in enumerate(body) if l.startswith(("global_load_dwordx2", "buffer_load_dwordx2")) and l in loop][2]
        dst = check_isa.vregs(check_isa.split_operands(body[idx])[1][0])[0]
        assert any("v_mov_b32" in p for p in check_isa.check_walk_loop(name, body[:idx + 1] + ["v_mov_b32_e32 v250, v%d" % dst] + body[idx + 1:]))
        assert any("expected (KR + D)" in p for p in check_isa.check_walk_loop(name, body[:idx + 1] + ["global_load_dwordx2 v[252:253], v0, s[0:1]"] + body[idx + 1:]))
        assert any("scratch_load" in p for p in check_isa.check_walk_loop(name, body[:idx + 1] + ["scratch_load_dword v250, off, s0"] + body[idx + 1:]))
        # the hand-downs of a multi-lane walk read the heads' mask from VCC, set once at the loop's head: a VCC writer among
        # the steps (here a VOPC compare in front of a hand-down) must be reported, and so must a head that does not set it
        dpp = [i for i, l in enumerate(body) if l.startswith("v_cndmask_b32_dpp") and l in loop]
        if dpp and "mac_walk3_kernel" in name:
            assert any("touches vcc" in p for p in check_isa.check_walk_loop(name, body[:dpp[3]] + ["v_cmp_eq_u32_e32 vcc, v250, v251"] + body[dpp[3]:]))
            assert any("touches vcc" in p for p in check_isa.check_walk_loop(name, body[:dpp[3]] + ["s_and_b64 vcc, exec, s[6:7]"] + body[dpp[3]:]))
            head = next(i for i, l in enumerate(body) if l.startswith("s_setprio"))
            assert body[head + 1].startswith("s_mov_b64 vcc")
            assert any("does not set it" in p for p in check_isa.check_walk_loop(name, body[:head + 1] + body[head + 2:]))
