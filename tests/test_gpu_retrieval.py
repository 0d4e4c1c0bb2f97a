"""GPU tests of the two-tower retrieval rows, mirroring the reference's
tests/keras/test_factorized_top_k.py and tests/keras/test_sbcnm.py, checked against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_semantics as O
from oracle import torch_ref as T

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_take_long_axis_and_exclude_known_answers():
    from deep_recommenders_amd.keras.models.retrieval import factorized_top_k as ftk
    g = G["take_long_axis"]                       # tests/keras/test_factorized_top_k.py:17-23
    out = ftk._take_long_axis(np.array(g["arr"], np.float32), np.array(g["indices"]))
    np.testing.assert_allclose(out.cpu().numpy(), np.array(g["expected"]), rtol=1e-6)
    g = G["exclude"]                              # :25-34
    x, y = ftk._exclude(np.array(g["scores"], np.float32), np.array(g["identifiers"]), np.array(g["exclude"]), g["k"])
    np.testing.assert_allclose(x.cpu().numpy(), np.array(g["expected_scores"]), rtol=1e-6)
    assert y.cpu().tolist() == g["expected_ids"]


@pytest.mark.parametrize("layer", ["Streaming", "BruteForce", None])
def test_factorized_topk_metrics(layer):
    # tests/keras/test_factorized_top_k.py:86-130
    from deep_recommenders_amd.keras.models.retrieval import factorized_top_k as ftk
    from deep_recommenders_amd.keras.models.retrieval import FactorizedTopK
    rng = np.random.RandomState(42)
    nc, nq, d = 100, 10, 4
    candidates = rng.normal(size=(nc, d)).astype(np.float32)
    queries = rng.normal(size=(nq, d)).astype(np.float32)
    true_candidates = rng.normal(size=(nq, d)).astype(np.float32)
    positive_scores = (queries * true_candidates).sum(axis=1, keepdims=True)
    all_scores = np.concatenate([positive_scores, queries @ candidates.T], axis=1)
    ks = [1, 5, 10, 50]
    batched = [candidates[i:i + 32] for i in range(0, nc, 32)]
    cands = batched if layer is None else getattr(ftk, layer)().index(batched)
    metric = FactorizedTopK(candidates=cands, metrics=[ftk.TopKCategoricalAccuracy(k=x, name=f"top_{x}_categorical_accuracy")
                                                       for x in ks], k=max(ks))
    metric.update_state(query_embeddings=queries, true_candidate_embeddings=true_candidates)
    for k, value in zip(ks, metric.result()):
        want = O.in_top_k(np.zeros(nq, np.int64), all_scores, k).mean()
        assert abs(value - want) < 1e-9
    metric.reset_states()
    assert metric.result() == [0.0] * len(ks)


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
def test_topk_scan_index_parity_against_fp64_brute_force(split):
    """The register-split exact scan (dr_topk_mips: dense first chunk + filtered chunks + running lists; factorized_top_k.py:201-233) at a
    size that takes that path (512 queries x 200 000 items x 128, k = 100), in both operand splits, against the float64 brute force
    with the reference's tie rule (equal scores -> lower index first, [TF] B13): the reported index at every rank either IS the
    brute force's, or the two candidates' float64 scores differ by less than the products' rounding (2e-6 of the largest score: a
    near-tie that no fp32 scan can order); exact duplicates of a row must come out in index order; the reported scores are the
    float64 scores of the reported indices to the same tolerance."""
    from deep_recommenders_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    Bq, N, Dq, k = 512, 200_000, 128, 100
    q = torch.randn((Bq, Dq), device="cuda", generator=g) / Dq ** 0.5
    corpus = torch.randn((N, Dq), device="cuda", generator=g) / Dq ** 0.5
    corpus[150_000] = corpus[77]                       # exact duplicates, one of them in a later chunk of the scan
    corpus[40_001] = corpus[77]
    corpus[77] *= 3.0; corpus[150_000] *= 3.0; corpus[40_001] *= 3.0       # ... and large enough to be in many rows' top-k
    prev = ops.set_gemm_split(split)
    try:
        s, idx = ops.topk_mips(q, corpus, k)
    finally:
        ops.set_gemm_split(prev)
    S = q.double() @ corpus.double().T                 # checker only: float64 brute force
    order = torch.sort(S, dim=1, descending=True, stable=True)
    want_i, want_s = order.indices[:, :k], order.values[:, :k]
    got_s64 = torch.gather(S, 1, idx)
    tol = 2e-6 * float(S.abs().max())
    assert float((got_s64 - want_s).abs().max()) <= tol, "a reported candidate is not (within rounding) the brute force's at its rank"
    assert float((s.double() - got_s64).abs().max()) <= tol
    same = float((idx == want_i).double().mean())
    # near-ties are rare but not absent: the top-100 scores of a row span ~0.09 (mean gap 9e-4) and two candidates closer than the
    # products' rounding (~3e-6) may swap -- ~0.3 % of adjacent pairs at this size (measured: 0.15 % of the positions differ)
    assert same >= 0.99, same
    assert int((torch.sort(idx, dim=1).values[:, 1:] == torch.sort(idx, dim=1).values[:, :-1]).sum()) == 0
    pos = {v: (idx == v).double().argmax(1) for v in (77, 40_001, 150_000)}
    has = (idx == 77).any(1) & (idx == 40_001).any(1) & (idx == 150_000).any(1)
    assert int(has.sum()) > 50
    assert bool((pos[77][has] < pos[40_001][has]).all()) and bool((pos[40_001][has] < pos[150_000][has]).all())


def test_topk_mips_matches_oracle_large_and_ties():
    from deep_recommenders_amd import ops
    rng = np.random.default_rng(5)
    Bq, N, D, k = 77, 5003, 32, 100
    q = rng.standard_normal((Bq, D)).astype(np.float32)
    cand = rng.standard_normal((N, D)).astype(np.float32)
    cand[1234] = cand[17]            # exact duplicate rows: ties must resolve to the lower index
    cand[4000] = cand[17]
    # small workspace -> several chunks (exercises the running-list continuation)
    ws = torch.empty(Bq * 512, dtype=torch.float32, device="cuda")
    s, idx = ops.topk_mips(torch.tensor(q).cuda(), torch.tensor(cand).cuda(), k, workspace=ws)
    ws_scores = q.astype(np.float64) @ cand.T.astype(np.float64)
    want_s, want_i = O.top_k((q @ cand.T).astype(np.float32), k)
    got_i = idx.cpu().numpy()
    got_s = s.cpu().numpy()
    # scores equal to fp32 round-off; index sets equal except where fp32 near-ties reorder -> compare via scores
    np.testing.assert_allclose(got_s, np.take_along_axis(ws_scores, got_i, 1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got_s, want_s, rtol=1e-5, atol=1e-5)
    assert np.all(np.diff(got_s, axis=1) <= 0)
    for r in range(Bq):
        row = got_i[r].tolist()
        if 17 in row and 1234 in row:
            assert row.index(17) < row.index(1234)
        if 1234 in row and 4000 in row:
            assert row.index(1234) < row.index(4000)
    # k > N on a fresh search -> the reference's ValueError
    with pytest.raises(ValueError):
        ops.topk_mips(torch.tensor(q).cuda(), torch.tensor(cand[:5]).cuda(), 10)


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
def test_topk_ties_at_the_kth_boundary_do_not_depend_on_arrival_order(split):
    """tf.math.top_k's rule (equal scores -> lower index first, [TF] B13) AT THE k-TH BOUNDARY of the filtered scan: 40 exact copies of
    one item, scattered over the later chunks of the corpus, outscore everything else for every query, and k = 10 -- the answer is the 10
    copies with the lowest indices, in index order, for every row.  The filtering epilogue appends a chunk's survivors to a row's
    candidate list with atomics (arbitrary order), so the list's gate has to be the TOTAL order (score, then index): with a gate on
    the score alone, a copy that arrives after the list has filled with later copies is refused although it precedes them."""
    from deep_recommenders_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    Bq, N, D, k = 512, 200_000, 128, 10
    q = 0.1 * torch.randn((Bq, D), device="cuda", generator=g)
    q[:, 0] = 1.0
    c = torch.randn((N, D), device="cuda", generator=g) / D ** 0.5
    u = torch.zeros(D, device="cuda")
    u[0] = 10.0
    copies = [40_000 + 3989 * t for t in range(40)]                    # 40 000 .. 195 571: behind the dense first chunk, several chunks
    c[copies] = u
    prev = ops.set_gemm_split(split)
    try:
        for _ in range(3):                                              # (the arrival order differs from run to run)
            s, i = ops.topk_mips(q, c, k)
            got = i.cpu().numpy()
            assert (got == np.asarray(copies[:k])[None, :]).all(), got[(got != np.asarray(copies[:k])[None, :]).any(1)][:4]
            assert bool((s == 10.0).all())
    finally:
        ops.set_gemm_split(prev)


@pytest.mark.parametrize("Bq,N,D,k", [(512, 200_000, 128, 100), (77, 5003, 20, 10), (64, 70_001, 64, 50)])
def test_topk_index_gives_the_same_bits_and_follows_the_corpus(Bq, N, D, k):
    """dr_topk_index_build / dr_topk_mips_indexed (BruteForce.index, factorized_top_k.py:275-297: the candidates are handed over once):
    the scan over a pre-split corpus returns exactly what the scan over the corpus returns -- scores and indices, bit for bit, in both
    operand splits (bf16x3 ignores the index) -- and an index whose corpus tensor was written in place is rebuilt on its next use."""
    from deep_recommenders_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    q = torch.randn((Bq, D), device="cuda", generator=g) / D ** 0.5
    corpus = torch.randn((N, D), device="cuda", generator=g) / D ** 0.5
    corpus[N - 3] = corpus[11]                                       # a duplicate across chunks: ties must fall the same way
    index = ops.TopKIndex(corpus)
    assert index.buf is not None
    for split in ("f16x2", "bf16x3"):
        prev = ops.set_gemm_split(split)
        try:
            s0, i0 = ops.topk_mips(q, corpus, k)
            s1, i1 = ops.topk_mips(q, index, k)
        finally:
            ops.set_gemm_split(prev)
        assert torch.equal(s0, s1) and torch.equal(i0, i1), split
    corpus[5] *= 50.0                                                 # in place: the planes AND the corpus' amax record are stale now
    s0, i0 = ops.topk_mips(q, corpus, k)
    s1, i1 = ops.topk_mips(q, index, k)
    assert torch.equal(s0, s1) and torch.equal(i0, i1)
    assert int((i1 == 5).sum()) > 0


def test_rows_scale_modes():
    """dr_rows_scale: the `Faiss` index' row scalings (faiss.normalize_L2 with zero rows left alone, factorized_top_k.py:370-371 of the
    reference; the k-means centroid = member sum / count with empty clusters kept, :372) -- bit-exact against the same fp32 expressions."""
    from deep_recommenders_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    x = torch.randn((1000, 20), device="cuda", generator=g)
    x[7] = 0
    s = ops.rowdot(x, x).reshape(-1)
    n = ops.rows_scale(x, s, mode=1)
    want = torch.where(s[:, None] > 0, x / torch.sqrt(s)[:, None], x)
    assert torch.equal(n, want) and bool((n[7] == 0).all())
    assert float((n.norm(dim=1)[torch.arange(1000, device="cuda") != 7] - 1).abs().max()) < 1e-6
    cnt = torch.randint(0, 4, (1000,), device="cuda", generator=g).float()
    fb = torch.randn((1000, 20), device="cuda", generator=g)
    m = ops.rows_scale(x, cnt, mode=2, fallback=fb)
    assert torch.equal(m, torch.where(cnt[:, None] > 0, x / cnt[:, None].clamp(min=1.0), fb))
    assert torch.equal(ops.rows_scale(x, cnt, mode=0), x * cnt[:, None])


def test_topk_index_whose_planes_are_further_apart_than_2gb_still_scans_both_terms():
    """the GEMM kernels reach the second fp16 plane of the corpus through a 32-bit buffer offset; an index of N = 600 000
    items x D = 512 puts it 2.3 GB behind the first -- past the resource's range, where loads return zeros and the scan would run on
    the h terms alone, silently (1e-3 score errors, ties lost).  dr_topk_mips_indexed must then split per chunk like the un-indexed scan:
    same bits as dr_topk_mips, and both against an fp64 brute force."""
    from deep_recommenders_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    Bq, N, D, k = 64, 600_000, 512, 20
    q = torch.randn((Bq, D), device="cuda", generator=g) / D ** 0.5
    corpus = torch.randn((N, D), device="cuda", generator=g) / D ** 0.5
    corpus[N - 7] = corpus[123]                                      # an exact tie across the whole corpus
    index = ops.TopKIndex(corpus)
    s0, i0 = ops.topk_mips(q, corpus, k)
    s1, i1 = ops.topk_mips(q, index, k)
    assert torch.equal(s0, s1) and torch.equal(i0, i1)
    ref = torch.empty((Bq, N), dtype=torch.float64, device="cuda")
    for c0 in range(0, N, 100_000):
        ref[:, c0:c0 + 100_000] = q.double() @ corpus[c0:c0 + 100_000].double().t()
    rs, ri = torch.sort(ref, dim=1, descending=True, stable=True)
    want = rs[:, :k]
    assert float((s1.double() - want).abs().max()) <= 3e-6 * float(want.abs().max())       # fp32-grade scores: BOTH terms were multiplied
    # index sets agree wherever the fp64 scores are further apart than that
    gap_ok = (rs[:, k - 1] - rs[:, k]) > 1e-5
    same = torch.sort(i1, dim=1).values == torch.sort(ri[:, :k], dim=1).values
    assert bool(same[gap_ok].all())


@pytest.mark.parametrize("N,nlist", [(100_003, 1024), (5000, 1), (70_000, 8192), (300, 7)])
def test_ivf_build_lists_is_a_stable_counting_sort(N, nlist):
    """dr_ivf_build_lists == np.argsort(assign, kind="stable") + the list boundaries (bit-exact); assignments outside [0, nlist) drop
    their vector (the grouping half of faiss' index.add, keras/models/retrieval/factorized_top_k.py:374-391 of the reference)."""
    from deep_recommenders_amd import ops
    rng = np.random.default_rng(5)
    a = rng.integers(0, nlist, size=N)
    if N > 1000:
        a[rng.integers(0, N, size=17)] = -1
        a[rng.integers(0, N, size=5)] = nlist
    order, ls = ops.ivf_build_lists(torch.from_numpy(a).cuda(), nlist)
    ok = (a >= 0) & (a < nlist)
    want_order = np.argsort(np.where(ok, a, nlist), kind="stable")[:int(ok.sum())]
    want_ls = np.concatenate([[0], np.cumsum(np.bincount(a[ok], minlength=nlist))])
    np.testing.assert_array_equal(ls.cpu().numpy(), want_ls)
    np.testing.assert_array_equal(order.cpu().numpy()[:int(ok.sum())], want_order)


def test_faiss_ivf_flat_index():
    """the `Faiss` (IVF-Flat, inner product) index.  Pinned like the reference pins it
    (tests/keras/test_factorized_top_k.py:36-130): exact top-k with nlist=1 on 100 items, save/load self-consistency; plus
    the IVF semantics against the oracle given the index's own centroids / assignments, exactness at nprobe == nlist,
    L2-normalisation, identifiers, and the error messages."""
    from deep_recommenders_amd.keras.models.retrieval import factorized_top_k as ftk
    rng = np.random.default_rng(42)
    # --- exact with one list (reference test shape: 100 candidates, D=4, k=10) ---
    cand = rng.standard_normal((100, 4)).astype(np.float32)
    q = rng.standard_normal((16, 4)).astype(np.float32)
    idx = ftk.Faiss(k=10, nlist=1).index(torch.tensor(cand))
    s, i = idx(torch.tensor(q))
    want_s, want_i = O.brute_force_top_k(q, cand, None, 10)
    np.testing.assert_allclose(s.cpu().numpy(), want_s, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(i.cpu().numpy(), want_i)
    # --- IVF semantics: nlist=16, nprobe=4 ---
    N, D, Bq, k = 5000, 32, 64, 20
    cand = rng.standard_normal((N, D)).astype(np.float32)
    ids = rng.permutation(10 * N)[:N].astype(np.int64)
    q = rng.standard_normal((Bq, D)).astype(np.float32)
    ivf = ftk.Faiss(k=k, nlist=16, nprobe=4).index(torch.tensor(cand), torch.tensor(ids))
    s, i = ivf(torch.tensor(q))
    cen = ivf._centroids.cpu().numpy()
    asg = ivf._assignments.cpu().numpy()
    assert cen.shape == (16, D) and np.bincount(asg, minlength=16).min() > 0
    np.testing.assert_array_equal(asg, np.argmax(cand.astype(np.float64) @ cen.astype(np.float64).T, axis=1))
    want_s, want_i = O.ivf_flat_search(q, cand, ids, cen, asg, 4, k)
    np.testing.assert_allclose(s.cpu().numpy(), want_s, rtol=1e-5, atol=1e-5)
    assert (i.cpu().numpy() == want_i).mean() > 0.999                      # fp32 near-ties may swap neighbours
    exact_s, _ = O.brute_force_top_k(q, cand, ids, k)
    recall = np.mean([len(set(i.cpu().numpy()[r]) & set(O.brute_force_top_k(q[r:r + 1], cand, ids, k)[1][0])) / k for r in range(Bq)])
    assert 0.3 < recall <= 1.0                                               # 4 of 16 lists of isotropic data: approximate
    # --- probing every list is exact ---
    s_all, i_all = ftk.Faiss(k=k, nlist=16, nprobe=16).index(torch.tensor(cand), torch.tensor(ids))(torch.tensor(q))
    np.testing.assert_allclose(s_all.cpu().numpy(), exact_s, rtol=1e-5, atol=1e-5)
    # --- save / load self-consistency (state travels with the module) ---
    clone = ftk.Faiss(k=k, nlist=16, nprobe=4)
    state = ivf.state_dict()
    for name in ("_centroids", "_packed", "_packed_ids", "_blk_off"):
        setattr(clone, name, state[name].clone())
    s2, i2 = clone(torch.tensor(q))
    assert torch.equal(s, s2) and torch.equal(i, i2)
    # --- cosine (normalize=True) and k override ---
    cos = ftk.Faiss(k=5, nlist=1, normalize=True).index(torch.tensor(cand))
    sc, ic = cos(torch.tensor(q), k=3)
    cn = cand / np.linalg.norm(cand, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    w_s, w_i = O.brute_force_top_k(qn, cn, None, 3)
    np.testing.assert_allclose(sc.cpu().numpy(), w_s, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(ic.cpu().numpy(), w_i)
    # --- fewer members than k in the probed lists -> (-inf, -1) tail, like faiss' -1 labels ---
    tiny = ftk.Faiss(k=8, nlist=4, nprobe=1).index(torch.tensor(cand[:12]))
    st, it = tiny(torch.tensor(q[:4]))
    assert (it.cpu().numpy() == -1).any() and np.isneginf(st.cpu().numpy()[it.cpu().numpy() == -1]).all()
    # --- error surface ---
    with pytest.raises(ValueError, match="must be called first"):
        ftk.Faiss()(torch.tensor(q))
    with pytest.raises(ValueError, match="ndim should be 2"):
        ftk.Faiss().index(torch.zeros(5))
    with pytest.raises(ValueError, match="Queries must be a tensor"):
        idx({"a": torch.tensor(q)})


def test_streaming_and_bruteforce_api():
    from deep_recommenders_amd.keras.models.retrieval import factorized_top_k as ftk
    rng = np.random.default_rng(6)
    cand = rng.standard_normal((300, 8)).astype(np.float32)
    q = rng.standard_normal((9, 8)).astype(np.float32)
    ids = (np.arange(300) * 7 + 3).astype(np.int64)
    with pytest.raises(ValueError):
        ftk.BruteForce()(q)                                   # "`index` method must be called first" (:323-325)
    with pytest.raises(ValueError):
        ftk.Streaming()(q)
    with pytest.raises(ValueError):
        ftk.BruteForce().index(np.zeros((3, 4, 5), np.float32))   # candidates ndim != 2 (:288-290)
    bf = ftk.BruteForce(k=10).index(cand, ids)
    st = ftk.Streaming(k=10).index([cand[i:i + 64] for i in range(0, 300, 64)], [ids[i:i + 64] for i in range(0, 300, 64)])
    want_s, want_i = O.brute_force_top_k(q, cand, ids, k=10)
    for layer in (bf, st):
        s, i = layer(q)
        np.testing.assert_allclose(s.cpu().numpy(), want_s, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(i.cpu().numpy(), want_i)
    # incomplete batches: first batch smaller than k (handle_incomplete_batches=True, :206-209)
    st2 = ftk.Streaming(k=10).index([cand[:4], cand[4:300]])
    s, i = st2(q)
    np.testing.assert_array_equal(i.cpu().numpy(), O.brute_force_top_k(q, cand, None, k=10)[1])
    with pytest.raises(ValueError):
        ftk.Streaming(k=10, handle_incomplete_batches=False).index([cand[:4], cand[4:]])(q)
    # query_with_exclusions (:111-129)
    excl = want_i[:, :2]
    s, i = bf.query_with_exclusions(q, excl, k=3)
    ws, wi = O.brute_force_top_k(q, cand, ids, k=5)
    es, ei = O.exclude(ws, wi, excl, 5)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)
    # state round trip: the index lives in buffers (:292-311)
    bf2 = ftk.BruteForce(k=10).index(np.zeros_like(cand), np.zeros_like(ids))
    bf2.load_state_dict(bf.state_dict())
    np.testing.assert_array_equal(bf2(q)[1].cpu().numpy(), want_i)


@pytest.mark.parametrize("h", [3, 5, 10, 15])
def test_hard_negative_mining(h):
    # tests/keras/test_sbcnm.py:16-41
    from deep_recommenders_amd.keras.models.retrieval import sbcnm
    rng = np.random.RandomState(42)
    logits = rng.uniform(size=(2, 20)).astype(np.float32)
    labels = rng.permutation(np.eye(2, 20).T).T.astype(np.float32)
    ol, olab = sbcnm.HardNegativeMining(h)(logits, labels)
    ol, olab = ol.cpu().numpy(), olab.cpu().numpy()
    assert ol.shape[-1] == h + 1
    np.testing.assert_allclose((ol * olab).sum(-1), (logits * labels).sum(-1), rtol=1e-6)
    logits2 = logits + labels * 1000.0
    ol2, _ = sbcnm.HardNegativeMining(h)(logits2, labels)
    np.testing.assert_allclose(np.sort(logits2, axis=1)[:, -h - 1:], np.sort(ol2.cpu().numpy()), rtol=1e-6)


def test_remove_accidental_negative_and_sampling_correction():
    # tests/keras/test_sbcnm.py:43-55
    from deep_recommenders_amd.keras.models.retrieval import sbcnm
    rng = np.random.RandomState(42)
    logits = rng.uniform(size=(2, 4)).astype(np.float32)
    labels = rng.permutation(np.eye(2, 4).T).T.astype(np.float32)
    identifiers = rng.randint(0, 3, size=4)
    out = sbcnm.RemoveAccidentalNegative()(logits, labels, identifiers).cpu().numpy()
    np.testing.assert_allclose((out * labels).sum(1), (logits * labels).sum(1), rtol=1e-6)
    np.testing.assert_allclose(out, O.remove_accidental_negative(logits, labels, identifiers), rtol=1e-6)
    p = rng.uniform(0.1, 0.9, size=4).astype(np.float32)
    out = sbcnm.SamplingProbabilityCorrection()(logits, p).cpu().numpy()
    np.testing.assert_allclose(out, O.sampling_probability_correction(logits, p), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("B,D", [(6, 4), (200, 32), (513, 128), (1024, 64), (2048, 128), (700, 20)])
def test_retrieval_loss_and_gradients(B, D, split):
    """(B >= 256 in the f16x2 split: both score passes on the register-split kernel with the LSE / softmax-gradient epilogues --
    513 and 700 have edge tiles in both directions, D = 20 a padded reduction; otherwise the fp32 MFMA kernel)"""
    from deep_recommenders_amd import ops
    from deep_recommenders_amd.keras.models.retrieval import sbcnm
    prev = ops.set_gemm_split(split)
    try:
        _retrieval_loss_and_gradients(B, D, sbcnm)
    finally:
        ops.set_gemm_split(prev)


def _retrieval_loss_and_gradients(B, D, sbcnm):
    rng = np.random.default_rng(8)
    q = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
    c = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, size=B).astype(np.float32)
    p = rng.uniform(0.05, 0.9, size=B).astype(np.float32)
    ids = rng.integers(0, max(2, B // 3), size=B)
    cases = [dict(), dict(temperature=0.5), dict(sample_weight=w), dict(candidate_sampling_probability=p),
             dict(candidate_ids=ids), dict(temperature=0.7, sample_weight=w, candidate_sampling_probability=p, candidate_ids=ids)]
    for kw in cases:
        temperature = kw.pop("temperature", None)
        task = sbcnm.Retrieval(temperature=temperature)
        tq = torch.tensor(q, device="cuda", requires_grad=True)
        tc = torch.tensor(c, device="cuda", requires_grad=True)
        loss = task(tq, tc, compute_metrics=False, **kw)
        want = O.retrieval_loss(q, c, sample_weight=kw.get("sample_weight"),
                                candidate_sampling_probability=kw.get("candidate_sampling_probability"),
                                candidate_ids=kw.get("candidate_ids"), temperature=temperature)
        assert abs(loss.item() - float(want)) <= 2e-5 * abs(float(want)) + 1e-4, (kw, loss.item(), want)
        loss.backward()
        oq = torch.tensor(q, dtype=torch.float64, requires_grad=True)
        oc = torch.tensor(c, dtype=torch.float64, requires_grad=True)
        lo = T.inbatch_softmax_loss(oq, oc, None if "sample_weight" not in kw else torch.tensor(w, dtype=torch.float64),
                                    None if "candidate_sampling_probability" not in kw else torch.tensor(p, dtype=torch.float64),
                                    None if "candidate_ids" not in kw else torch.tensor(ids), temperature)
        lo.backward()
        np.testing.assert_allclose(tq.grad.cpu().numpy(), oq.grad.numpy(), rtol=1e-3, atol=2e-5)
        np.testing.assert_allclose(tc.grad.cpu().numpy(), oc.grad.numpy(), rtol=1e-3, atol=2e-5)


def test_retrieval_with_metrics_and_hard_negatives():
    from deep_recommenders_amd.keras.models.retrieval import sbcnm, FactorizedTopK
    from deep_recommenders_amd.keras.models.retrieval import factorized_top_k as ftk
    rng = np.random.default_rng(9)
    B, D = 64, 16
    q = rng.standard_normal((B, D)).astype(np.float32)
    c = rng.standard_normal((B, D)).astype(np.float32)
    corpus = rng.standard_normal((500, D)).astype(np.float32)
    metric = FactorizedTopK(ftk.BruteForce().index(corpus), k=100)
    task = sbcnm.Retrieval(metrics=metric)
    loss = task(q, c)
    assert np.isfinite(loss.item()) and len(metric.result()) == 5 and all(0.0 <= v <= 1.0 for v in metric.result())
    assert task.factorized_metrics is metric
    # hard negatives: CCE over the positive + the h hardest negatives (sbcnm.py:145-151)
    h = 5
    loss_h = sbcnm.Retrieval(num_hard_negatives=h, temperature=0.5)(q, c, compute_metrics=False)
    want = O.retrieval_loss(q, c, temperature=0.5, num_hard_negatives=h)
    assert abs(loss_h.item() - float(want)) <= 1e-4 * abs(float(want))


def test_retrieval_hard_negatives_backward_matches_autograd_oracle():
    """Gradients of the num_hard_negatives branch (sbcnm.py:145-151) w.r.t. both towers, with sample weights, sampling
    probability correction, accidental-hit removal and temperature, against a float64 autograd restatement."""
    from deep_recommenders_amd.keras.models.retrieval import sbcnm
    rng = np.random.default_rng(19)
    B, D, h, temp = 96, 24, 7, 0.7
    q = rng.standard_normal((B, D)).astype(np.float32)
    c = rng.standard_normal((B, D)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, size=B).astype(np.float32)
    cp = rng.uniform(0.01, 0.9, size=B).astype(np.float32)
    ci = rng.integers(0, 40, size=B).astype(np.int64)               # duplicates -> accidental negatives
    tq = torch.tensor(q, device="cuda", requires_grad=True)
    tc = torch.tensor(c, device="cuda", requires_grad=True)
    task = sbcnm.Retrieval(num_hard_negatives=h, temperature=temp)
    loss = task(tq, tc, sample_weight=w, candidate_sampling_probability=cp, candidate_ids=ci, compute_metrics=False)
    loss.backward()
    # float64 restatement with torch autograd (CPU)
    oq = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    oc = torch.tensor(c, dtype=torch.float64, requires_grad=True)
    scores = oq @ oc.t()
    labels = torch.eye(B, dtype=torch.float64)
    scores = scores - torch.log(torch.tensor(cp, dtype=torch.float64))[None, :]              # :78-86
    ident = torch.tensor(ci).reshape(-1, 1)
    dup = (ident == ident.t()).double() - labels                                             # :66-73
    scores = scores + dup * float(sbcnm.MIN_FLOAT)
    boosted = (scores.detach().float() + labels.float() * float(sbcnm.MAX_FLOAT))          # selection in fp32 like the kernels
    idx = torch.topk(boosted, h + 1, dim=1).indices                                          # :41-44
    s_sel = torch.gather(scores, 1, idx) / temp                                              # :46-47,148-149
    l_sel = torch.gather(labels, 1, idx)
    row = (torch.logsumexp(s_sel, dim=1) * l_sel.sum(1) - (s_sel * l_sel).sum(1)) * torch.tensor(w, dtype=torch.float64)
    lo = row.sum()                                                                           # CCE(from_logits, SUM)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 1e-4 * abs(lo.item())
    np.testing.assert_allclose(tq.grad.cpu().numpy(), oq.grad.numpy(), rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), oc.grad.numpy(), rtol=1e-3, atol=5e-5)
    assert float(tq.grad.abs().sum()) > 0


def test_retrieval_user_supplied_loss():
    """Retrieval(loss=callable) (sbcnm.py:100-103): a custom loss on the explicit scores, gradients through the GEMMs.  With
    the default CCE/SUM written as a callable it must reproduce the fused path."""
    from deep_recommenders_amd.keras.models.retrieval import sbcnm
    rng = np.random.default_rng(23)
    B, D = 80, 16
    q = rng.standard_normal((B, D)).astype(np.float32)
    c = rng.standard_normal((B, D)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, size=B).astype(np.float32)

    def cce_sum(y_true, y_pred, sample_weight=None):
        row = torch.logsumexp(y_pred, dim=1) * y_true.sum(1) - (y_pred * y_true).sum(1)
        return (row * sample_weight).sum() if sample_weight is not None else row.sum()

    grads = []
    for loss_obj in (None, cce_sum):
        tq = torch.tensor(q, device="cuda", requires_grad=True)
        tc = torch.tensor(c, device="cuda", requires_grad=True)
        lo = sbcnm.Retrieval(loss=loss_obj, temperature=0.5)(tq, tc, sample_weight=w, compute_metrics=False)
        lo.backward()
        grads.append((lo.item(), tq.grad.cpu().numpy(), tc.grad.cpu().numpy()))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-5 * abs(grads[0][0])
    np.testing.assert_allclose(grads[0][1], grads[1][1], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(grads[0][2], grads[1][2], rtol=1e-4, atol=1e-5)
