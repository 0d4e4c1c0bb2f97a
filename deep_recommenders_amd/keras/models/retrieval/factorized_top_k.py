"""Top-K retrieval indices + the FactorizedTopK metric — same surface as the reference's
keras/models/retrieval/factorized_top_k.py (`_take_long_axis` :26-41, `_exclude` :44-67, `TopK` :70-136,
`Streaming` :139-260, `BruteForce` :263-334, `Faiss` :337-461, `FactorizedTopK` :464-522), on the K10 kernels."""
import abc
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from deep_recommenders_amd import ops


def _dev(t, dtype=None):
    t = torch.as_tensor(t)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda() if not t.is_cuda else t


def _take_long_axis(arr, indices):
    """arr [B, C], indices [B, K] -> arr[b, indices[b, k]]  (factorized_top_k.py:26-41)"""
    arr = _dev(arr)
    if arr.dtype not in (torch.float32, torch.int64):
        arr = arr.to(torch.float32 if arr.is_floating_point() else torch.int64)
    return ops.take_along_rows(arr, _dev(indices, torch.int64))


def _exclude(scores, identifiers, exclude, k):
    """Remove `exclude` identifiers from a top-k result (factorized_top_k.py:44-67, penalty 1e5 at :62)."""
    scores = _dev(scores, torch.float32)
    identifiers = _dev(identifiers, torch.int64)
    exclude = _dev(exclude, torch.int64)
    adjusted = ops.exclude_adjust(scores, identifiers, exclude)
    k = min(int(k), scores.shape[1])
    _, indices = ops.topk_select(adjusted, k)
    return _take_long_axis(scores, indices), _take_long_axis(identifiers, indices)


class TopK(nn.Module, abc.ABC):
    """TopK layer interface: `index(candidates, identifiers=None) -> self`, `call(queries, k=None) -> (scores, ids)`."""

    def __init__(self, k: int, *args, **kwargs):
        super().__init__()
        self._k = k

    @abc.abstractmethod
    def index(self, candidates, identifiers=None) -> "TopK":
        raise NotImplementedError("Implementers must provide `index` method.")

    @abc.abstractmethod
    def call(self, queries, k: Optional[int] = None, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError()

    def forward(self, queries, k: Optional[int] = None, **kwargs):
        return self.call(queries, k=k, **kwargs)

    def query_with_exclusions(self, queries, exclusions, k: Optional[int] = None):
        """factorized_top_k.py:111-129"""
        k = k if k is not None else self._k
        exclusions = _dev(exclusions, torch.int64)
        adjusted_k = k + exclusions.shape[1]
        scores, identifiers = self(queries=queries, k=adjusted_k)
        return _exclude(scores, identifiers, exclusions, adjusted_k)


def _batches(x) -> List[torch.Tensor]:
    """A 'dataset' here is any iterable of [n_i, D] batches (the reference uses tf.data.Dataset.batch)."""
    if isinstance(x, torch.Tensor):
        return [x]
    if hasattr(x, "shape") and not isinstance(x, (list, tuple)):
        return [torch.as_tensor(x)]
    return [torch.as_tensor(b) for b in x]


class Streaming(TopK):
    """Retrieves top k scoring items and identifiers from a large (batched) candidate set (:139-260):
    per candidate batch a top-k (map), merged into the running top-k (reduce) — here each batch is folded
    straight into the running per-query list by the K10 kernels."""

    def __init__(self, k: int = 10, query_model=None, handle_incomplete_batches: bool = True, num_parallel_calls=None,
                 sorted_order: bool = True, *args, **kwargs):
        super().__init__(k, *args, **kwargs)
        self._query_model = query_model
        self._handle_incomplete_batches = handle_incomplete_batches
        self._num_parallel_calls = num_parallel_calls
        self._sorted_order = sorted_order
        self._candidates = None
        self._identifiers = None

    def index(self, candidates, identifiers=None, **kwargs) -> "Streaming":
        self._candidates = candidates
        self._identifiers = identifiers
        return self

    def call(self, queries, k: Optional[int] = None, **kwargs):
        k = k if k is not None else self._k
        if self._candidates is None:
            raise ValueError("The `index` method must be called first to "
                             "create the retrieval index.")                       # :191-193
        if self._query_model is not None:
            queries = self._query_model(queries)
        queries = _dev(queries, torch.float32)
        cand_batches = _batches(self._candidates)
        id_batches = _batches(self._identifiers) if self._identifiers is not None else None
        total = sum(int(b.shape[0]) for b in cand_batches)
        if not self._handle_incomplete_batches:
            for b in cand_batches:
                if b.shape[0] < k:                                                 # :13-23,256
                    raise ValueError("Tried to retrieve k={k} top items, but candidate batch too small."
                                     "To resolve this, 1. increase batch-size, 2. set `drop_remainder`=True, "
                                     "3. set `handle_incomplete_batches`=True in constructor.".format(k=k))
        k_eff = min(k, total)
        state = ops.topk_init(queries.shape[0], k_eff, queries.device)      # initial_state: empty (:239-240)
        counter = 0                                                         # counter-based row ids (:244-254)
        for b in cand_batches:
            b = _dev(b, torch.float32)
            ops.topk_mips(queries, b, k_eff, index_base=counter, init=False, state=state)
            counter += b.shape[0]
        scores, index = state
        if id_batches is not None:
            ids = torch.cat([_dev(i).reshape(-1) for i in id_batches]).to(torch.int64)
            return scores, ops.gather_i64(ids, index)
        return scores, index


class BruteForce(TopK):
    """Brute-force retrieval (:263-334): scores = queries @ candidates^T, top-k, gather identifiers."""

    def __init__(self, k: int = 10, query_model=None, *args, **kwargs):
        super().__init__(k, *args, **kwargs)
        self._query_model = query_model
        # non-trainable state so the index serialises with the module (:292-311)
        self.register_buffer("_candidates", None)
        self.register_buffer("_identifiers", None)

    def index(self, candidates, identifiers=None) -> "BruteForce":
        cand = torch.cat([_dev(b, torch.float32) for b in _batches(candidates)], dim=0)
        if cand.dim() != 2:
            raise ValueError("`candidates` ndim should be 2. "
                             "Got `ndim` = {}".format(cand.dim()))                  # :288-290
        if identifiers is None:
            ids = torch.arange(cand.shape[0], device=cand.device, dtype=torch.int64)
        else:
            ids = torch.cat([_dev(b).reshape(-1) for b in _batches(identifiers)]).to(torch.int64)
        self._candidates = cand.contiguous()
        self._identifiers = ids.contiguous()
        # the corpus side of the scan (amax record + fp16 planes of the candidates), derived here once instead of on every call
        self._index = ops.TopKIndex(self._candidates) if self._candidates.is_cuda else None
        return self

    def call(self, queries, k: Optional[int] = None, **kwargs):
        k = k if k is not None else self._k
        if self._candidates is None:
            raise ValueError("The `index` method must be called first to "
                             "create the retrieval index.")                       # :323-325
        if self._query_model is not None:
            queries = self._query_model(queries)
        queries = _dev(queries, torch.float32)
        scores, index = ops.topk_mips(queries, self._index if getattr(self, "_index", None) is not None else self._candidates, k)   # :330-332
        return scores, ops.gather_i64(self._identifiers, index)                    # :334


class Faiss(TopK):
    """IVF-Flat retrieval index for a factorized retrieval model (:337-461) -- same constructor (`k`, `query_model`, `nlist`,
    `nprobe`, `normalize`), `index(candidates, identifiers=None)`, `call(queries, k=None)`.

    The reference delegates to faiss-cpu 1.6.3 (`IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT)`, :367-371),
    a third-party library that is neither in the reference tree nor in this image.  This is the same index family on the
    GPU (csrc/ivf.hip): k-means centroids trained on the candidates with inner-product assignment, every candidate stored
    in the list of its best centroid, a query scans the `nprobe` best lists exactly.  The trained centroids are this
    implementation's own (seeded, `niter` Lloyd iterations from a random subset), so for `nprobe < nlist` the approximate
    results are those of THIS index, not bit-identical to faiss'; with `nlist == 1` or `nprobe == nlist` the search is exact
    (the only behaviour the reference's tests pin: tests/keras/test_factorized_top_k.py:86-130)."""

    def __init__(self, k: int = 10, query_model=None, nlist: Optional[int] = 1, nprobe: Optional[int] = 1,
                 normalize: bool = False, niter: int = 10, seed: int = 1234, *args, **kwargs):
        super().__init__(k, *args, **kwargs)
        self._query_model = query_model
        self._nlist = int(nlist)
        self._nprobe = int(nprobe)
        self._normalize = normalize
        self._niter = int(niter)
        self._seed = int(seed)
        for name in ("_centroids", "_packed", "_packed_ids", "_blk_off", "_identifiers"):
            self.register_buffer(name, None)
        self._assignments = None         # list of every candidate (kept for inspection / tests)

    @staticmethod
    def _normalize_L2(x):
        x = x.contiguous()
        return ops.rows_scale(x, ops.rowdot(x, x), mode=1)                        # x / |x|; faiss.normalize_L2 leaves zero rows alone

    def _assign(self, x, centroids):
        _, best = ops.topk_mips(x, centroids, 1)                                  # coarse quantizer = exact inner product
        return best.reshape(-1)

    def _train(self, cand):
        """k-means with inner-product assignment (what `index.train(candidates)` does for an IP index, :372)."""
        N, D = cand.shape
        if N < self._nlist:
            raise ValueError("Number of training points ({}) should be at least as large as number of clusters ({})".format(
                N, self._nlist))
        g = torch.Generator(device=cand.device)
        g.manual_seed(self._seed)
        centroids = cand[torch.randperm(N, device=cand.device, generator=g)[:self._nlist]].clone()
        # the cluster sizes ride along with the sums in the scatter-add's first-order channel (a ones vector as the per-row scalar):
        # one launch per iteration; the list build -- hist + scan + scatter -- is only needed once, by index() (fp32 counts are exact
        # to 2^24 members per cluster)
        ones = torch.ones(N, dtype=torch.float32, device=cand.device)
        for _ in range(1 if self._nlist == 1 else self._niter):
            # (one list: every vector belongs to it -- its centroid is the mean, through the same two launches)
            a = self._assign(cand, centroids) if self._nlist > 1 else torch.zeros(N, dtype=torch.int64, device=cand.device)
            sums = torch.zeros((self._nlist, D), dtype=torch.float32, device=cand.device)
            counts = torch.zeros(self._nlist, dtype=torch.float32, device=cand.device)
            ops.rows_scatter_add(a, cand, ones, 1.0, sums, counts)
            centroids = ops.rows_scale(sums, counts, mode=2, fallback=centroids.contiguous())   # an empty cluster keeps its centroid
        return centroids

    def index(self, candidates, identifiers=None) -> "Faiss":
        cand = torch.cat([_dev(b, torch.float32) for b in _batches(candidates)], dim=0)
        if cand.dim() != 2:
            raise ValueError("`candidates` ndim should be 2. "
                             "Got `ndim` = {}".format(cand.dim()))                   # :409-411
        N = cand.shape[0]
        ids = None
        self._identifiers = None
        if identifiers is not None:
            idt = torch.cat([_dev(b).reshape(-1) for b in _batches(identifiers)])
            if idt.dtype in (torch.int8, torch.int16, torch.int32, torch.int64):
                ids = idt.to(torch.int64).contiguous()                                # add_with_ids (:387)
            else:
                self._identifiers = idt                                               # gathered after the search (:413-427,461)
        if self._normalize is True:
            cand = self._normalize_L2(cand)                                           # :370-371
        cand = cand.contiguous()
        self._centroids = self._train(cand).contiguous()
        assign = self._assign(cand, self._centroids)
        self._assignments = assign
        order, list_start = ops.ivf_build_lists(assign, self._nlist)            # stable counting sort by list (dr_ivf_build_lists)
        self._packed, self._packed_ids, self._blk_off = ops.ivf_pack(cand, order, list_start, ids)
        return self

    def call(self, queries, k: Optional[int] = None, **kwargs):
        k = k if k is not None else self._k
        if self._packed is None:
            raise ValueError("The `index` method must be called first to "
                             "create the retrieval index.")                       # :437-439
        if self._query_model is not None:
            queries = self._query_model(queries)
        if isinstance(queries, dict) or not (isinstance(queries, torch.Tensor) or hasattr(queries, "shape")):
            raise ValueError("Queries must be a tensor, got {}.".format(type(queries)))        # :444-445
        queries = _dev(queries, torch.float32)
        if self._normalize is True:
            queries = self._normalize_L2(queries)                                  # :450-451
        nprobe = max(1, min(self._nprobe, self._nlist))                             # searcher.nprobe (:453)
        _, probes = ops.topk_mips(queries.contiguous(), self._centroids, nprobe)
        distances, indices = ops.ivf_scan(queries.contiguous(), probes, self._blk_off, self._packed, self._packed_ids, int(k))
        if self._identifiers is None:
            return distances, indices
        return distances, self._identifiers[indices.clamp(min=0)]                   # :461


class TopKCategoricalAccuracy:
    """[TF] tf.keras.metrics.TopKCategoricalAccuracy(k) restricted to what FactorizedTopK feeds it
    (y_true = [1, 0, ...], y_pred = [positive, top-k...]): running mean of in_top_k (App. B14)."""

    def __init__(self, k: int = 5, name: Optional[str] = None):
        self.k = int(k)
        self.name = name or "top_k_categorical_accuracy"
        self.reset_states()

    def reset_states(self):
        self.hits = 0
        self.count = 0

    def result(self) -> float:
        return self.hits / self.count if self.count else 0.0


class FactorizedTopK(nn.Module):
    """Metric for a retrieval model (:464-522): top-{1,5,10,50,100} categorical accuracy of the true candidate
    against the top-k retrieved from `candidates` (a TopK layer or an iterable of candidate batches)."""

    def __init__(self, candidates, metrics: Optional[Sequence[TopKCategoricalAccuracy]] = None, k: int = 100,
                 name: str = "factorized_top_k", **kwargs):
        super().__init__()
        self.name = name
        if metrics is None:
            metrics = [TopKCategoricalAccuracy(k=n, name=f"{self.name}/top_{n}_categorical_accuracy")
                       for n in [1, 5, 10, 50, 100]]                                # :475-480
        if not isinstance(candidates, TopK):
            candidates = Streaming(k=k).index(candidates)                           # :482-483
        self._candidates = candidates
        self._metrics = list(metrics)
        self._k = k

    @property
    def metrics(self):
        return self._metrics

    def update_state(self, query_embeddings, true_candidate_embeddings) -> None:
        q = _dev(query_embeddings, torch.float32)
        c = _dev(true_candidate_embeddings, torch.float32)
        positive_scores = ops.rowdot(q, c)                                          # :494-495
        top_k_predictions, _ = self._candidates(q, k=self._k)                      # :497
        ks = torch.tensor([m.k for m in self._metrics], dtype=torch.int32, device=q.device)
        hits = torch.zeros(len(self._metrics), dtype=torch.int64, device=q.device)
        ops.topk_hits(positive_scores, top_k_predictions, ks, hits)                # :499-512 (in_top_k counting)
        h = hits.cpu().tolist()
        for m, v in zip(self._metrics, h):
            m.hits += int(v)
            m.count += q.shape[0]

    def reset_states(self) -> None:
        for metric in self.metrics:
            metric.reset_states()

    def result(self) -> List[float]:
        return [metric.result() for metric in self.metrics]
