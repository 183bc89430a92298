"""configs[1]'s job with every chunk different: 248 distinct 4096-site chunks (1 015 808 sites; 8 generated chunks x 31 variants each,
SynthChunk.variant) through ONE Engine - resident in HBM back to back, and from the host through Engine.run_stream - with the oracle on a
512-site sample spread over all 248 chunks.  The bench's timed region and `sustained` cycle 16 resident chunks; this is the run that would
show state carried from one chunk into the next (workspaces, key counters, the posterior's flags) if there were any.
Size-independent properties on top: the streamed results equal the resident pass's bit for bit, and re-running the first and the last chunk
after all the others reproduces them bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_a_million_distinct_sites_through_one_engine():
    import torch
    from concurrent.futures import ThreadPoolExecutor
    import bench
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda", 0)
    models = synthetic_models(bench.N_OUT, seed=0)
    lik, edges = lik_and_edges(likelihood_table(bench.N_OUT), bench.N_OUT)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    with ThreadPoolExecutor(max_workers=8) as ex:
        base = list(ex.map(lambda i: SynthChunk(bench.BATCH, seed=777 + i, start=100000 + i * 2000000), range(8)))
    first = eng.run_chunk(base[0].variant(0).arrays(), base[0].variant(0).site_pos)
    first = {k: first[k].cpu().numpy() for k in ("probs", "decision", "qual")}
    r = bench.sustained_distinct_leg(eng, base, models, lik, edges, 20, bench.BATCH, n_chunks=248, n_oracle=512)
    assert r["chunks"] == 248 and r["distinct_sites"] == 248 * bench.BATCH >= 1000000
    assert r["oracle"]["sites"] == 512 and r["oracle"]["chunks_sampled"] == 248
    assert r["oracle"]["max_abs_dP"] < 1e-4, r["oracle"]                    # north_star's tolerance
    assert r["oracle"]["decisions_equal_frac"] >= 0.99, r["oracle"]         # a genotype may turn on a probability 1e-6 from a bin edge
    assert r["run_stream"]["bit_equal_to_resident_pass"]
    again = eng.run_chunk(base[0].variant(0).arrays(), base[0].variant(0).site_pos)
    for k, v in first.items():
        np.testing.assert_array_equal(again[k].cpu().numpy(), v, err_msg=k)
    last = base[247 % 8].variant(247 // 8, shift=(247 // 8) * 40000000)
    a, b = eng.run_chunk(last.arrays(), last.site_pos), eng.run_chunk(last.arrays(), last.site_pos)
    for k in first:
        np.testing.assert_array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), err_msg=k)
    assert not np.array_equal(a["probs"].cpu().numpy(), first["probs"])     # and they really are different chunks
