"""The multi-GPU entries of include/svtav1_hip.h from C99 (tests/c_consumer/comm_consumer.c):
  * CPU: `comm_consumer plan` -- the C host's own view of svthip_shard_range / svthip_recon_slab_rows / svthip_recon_exchange_plan
    for the BASELINE shapes at 8 ranks equals what the ctypes mirror (svtav1_hip.sharded) sees, and every send has its receive;
  * GPU box (world 1): svthip_comm_create -> svthip_recon_exchange_dev on Y, Cb, Cr device planes allocated with the HIP C API ->
    the planes come back padded like PadRefAndSetFlags; svthip_me_gather_results_dev returns the rows."""
import os
import subprocess

import numpy as np
import pytest

from svtav1_hip import sharded

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CDIR = os.path.join(ROOT, "tests", "c_consumer")
EXE = os.path.join(CDIR, "comm_consumer")


def _build():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", CDIR, "comm_consumer"])


@pytest.mark.parametrize("case", [(1920, 1080, 160, 1, 8), (3840, 2160, 160, 2, 8), (856, 480, 96, 1, 3), (256, 64, 32, 1, 4)])
def test_c_host_plans_equal_the_ctypes_view(case):
    _build()
    w, h, origin, es, world = case
    r = subprocess.run([EXE, "plan"] + [str(v) for v in case], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    ranks, xfers = {}, {}
    for line in r.stdout.splitlines():
        f = line.split()
        if f[0] == "rank":
            ranks[int(f[1])] = tuple(int(v) for v in (f[3], f[4], f[6], f[7]))
        else:
            xfers.setdefault(int(f[1]), []).append(tuple(int(v) for v in f[2:]))
    n_sb = ((w + 63) // 64) * ((h + 63) // 64)
    pic = sharded.ReconPicture()
    pic.y = pic.cb = pic.cr = 16
    pic.stride_y, pic.stride_cb, pic.stride_cr = w + 2 * origin, w // 2 + origin, w // 2 + origin
    pic.width, pic.height, pic.origin_x, pic.origin_y, pic.sample_bytes = w, h, origin, origin, es
    covered = 0
    for rk in range(world):
        assert ranks[rk] == sharded.shard_range(n_sb, world, rk) + sharded.recon_slab_rows(h, world, rk)
        covered += ranks[rk][3]
        want = [(x.peer, x.plane, x.send, x.offset, x.bytes) for x in sharded.recon_exchange_plan(pic, world, rk)]
        assert xfers.get(rk, []) == want
    assert covered == h
    for a in range(world):
        for b in range(world):
            if a != b:
                assert [t[1:2] + t[3:] for t in xfers.get(a, []) if t[2] == 1 and t[0] == b] == [t[1:2] + t[3:] for t in xfers.get(b, []) if t[2] == 0 and t[0] == a]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1920, 1080, 160, 1), (640, 360, 96, 2)])
def test_c_host_exchange_world1(tmp_path, case):
    _build()
    w, h, origin, es = case
    dt = np.uint8 if es == 1 else np.uint16
    rng = np.random.default_rng(h)
    geo = [(w, h, origin), (w // 2, h // 2, origin // 2), (w // 2, h // 2, origin // 2)]
    truth = [rng.integers(0, 256 if es == 1 else 1024, (gh, gw)).astype(dt) for gw, gh, _ in geo]
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([w, h, origin, es], np.uint32).tobytes())
        for (gw, gh, gp), pic in zip(geo, truth):
            a = rng.integers(0, 200, (gh + 2 * gp, gw + 2 * gp)).astype(dt)   # junk borders
            a[gp:gp + gh, gp:gp + gw] = pic
            f.write(a.tobytes())
    r = subprocess.run([EXE, "exchange", inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = np.fromfile(outp, np.uint8)
    off = 0
    for (gw, gh, gp), pic in zip(geo, truth):
        n = (gh + 2 * gp) * (gw + 2 * gp) * es
        got = raw[off:off + n].view(dt).reshape(gh + 2 * gp, gw + 2 * gp)
        assert np.array_equal(got, np.pad(pic, gp, mode="edge"))
        off += n
    assert raw.size - off == 3 * 28 * 85 * 24
