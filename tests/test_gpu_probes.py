"""Hardware behaviour the kernels rely on, probed where the suite runs (GPU; skipped without hipcc).

  tools/probe/lds_xchg.hip  ds_wrxchg_rtn_b32: lanes of one instruction that hit the same LDS address are served in ascending lane order.
                            k_lz_candidates' byte-identity with oracle E rests on it (an out-of-order slot owner only drops a candidate: the
                            frame stays valid); the identity tests of this suite would turn flaky on a part where it does not hold - this
                            probe says so directly.
  tools/probe/lds_dma.hip   global_load_lds_dwordx4 lands in the issuing workgroup's LDS with two workgroups per CU and destinations beyond
                            64 KiB, and s_waitcnt vmcnt(N) covers it in issue order with a younger store (k_lz_walk's distance prefetch).
"""
import os, re, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build_and_run(name, tmp, timeout=240):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    exe = os.path.join(tmp, name)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tools", "probe", name + ".hip")])
    out = subprocess.run(["timeout", "-k", "5", str(timeout), exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


@pytest.mark.gpu
def test_lds_exchange_serves_same_address_lanes_in_lane_order(tmp_path):
    out = _build_and_run("lds_xchg", str(tmp_path))
    m = re.search(r"xchg order: (\d+) violations in (\d+) same-address successor checks", out)
    assert m, out
    assert int(m.group(1)) == 0 and int(m.group(2)) > 10_000_000, out


@pytest.mark.gpu
def test_lds_dma_lands_in_its_own_workgroup_and_in_issue_order(tmp_path):
    out = _build_and_run("lds_dma", str(tmp_path))
    lines = [l for l in out.splitlines() if l.startswith("wait vmcnt")]
    assert len(lines) == 8, out
    for l in lines:
        assert "no error" in l and l.rstrip().endswith("wrong pieces 0  foreign LDS words changed 0"), l
