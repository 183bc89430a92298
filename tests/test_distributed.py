"""World-size-2 test of the shard + gather path on CPU (gloo); the GPU path uses the same code over RCCL."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    # the collective helper must be importable without the HIP extension being exercised
    from clairs_to_amd.dist import shard_range, gather_site_outputs
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["PORT"],
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 1001                                    # odd: shards differ in size
    full = torch.arange(n * 8 * 2, dtype=torch.float32).reshape(n, 8, 2)
    lo, hi = shard_range(n, world, rank)
    out = gather_site_outputs(full[lo:hi].clone(), n)
    assert out.shape == full.shape and torch.equal(out, full), "gather is not bit-identical to the single-rank result"
    cover = [shard_range(n, world, r) for r in range(world)]
    assert cover[0][0] == 0 and cover[-1][1] == n and all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
    # the ragged form a real run uses (call_chunks --gather_outputs: ranks hold whole chunks, so their site counts differ freely)
    from clairs_to_amd.dist import gather_site_rows
    cut = [0, 700, n] if world == 2 else [0] + [n * (r + 1) // world for r in range(world)]
    for dtype in (torch.float32, torch.uint8, torch.int64, torch.float64):
        fullt = (full * 0 + torch.arange(n).reshape(n, 1, 1)).to(dtype)
        got, counts = gather_site_rows(fullt[cut[rank]:cut[rank + 1]].clone())
        assert counts == [cut[r + 1] - cut[r] for r in range(world)] and torch.equal(got, fullt)
    got, counts = gather_site_rows(torch.zeros((0, 3), dtype=torch.int32) if rank == 0 else torch.ones((5, 3), dtype=torch.int32))
    assert counts[0] == 0 and got.shape == (5 * (world - 1), 3)
    dist.destroy_process_group()
    print("rank", rank, "ok")
''') % ROOT


def test_shard_and_gather_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", PORT=port, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_shard_range_properties():
    from clairs_to_amd.dist import shard_range
    for n in (0, 1, 7, 4096, 10_000_000):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
