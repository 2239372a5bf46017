"""world_size-2 tests of the multi-process path (gloo, 127.0.0.1).

CPU: the distributed helpers bench.py and the sharded BA rely on (max over ranks, sequence sharding, the
all-reduce callback driven through its C function pointer).
GPU: the landmark-sharded local BA (fb_local_ba_sharded), two ranks sharing the one GPU of the box, against
the unsharded result."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world2(mode, timeout=240):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
    return outs


def test_distributed_helpers_world2_gloo():
    outs = _run_world2("helpers")
    assert all("helpers ok" in o for o in outs)


@pytest.mark.gpu
def test_sharded_local_ba_world2():
    outs = _run_world2("ba")
    assert all("flags_equal=True" in o and "identical_across_ranks=True" in o for o in outs)
