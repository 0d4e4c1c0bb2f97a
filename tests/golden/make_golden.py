"""Writes tests/golden/*.json — the known-answer vectors that pin the oracle.

Provenance of every vector (citations relative to /root/reference/, which is NOT readable on
the GPU box — hence committed fixtures):

* cross_kat        tests/keras/test_dcn.py:16-23   x0=[.1,.2,.3] x=[.4,.5,.6] W=ones b=0 -> [.55,.8,1.05]
* take_long_axis   tests/keras/test_factorized_top_k.py:17-23
* exclude          tests/keras/test_factorized_top_k.py:25-34
* farmhash         the reference holds NO hashed-id expectation (SURVEY.md §8c "parity unpinned"
                   at the reference level); these are the upstream TensorFlow / FarmHash vectors
                   listed in SURVEY.md §8c (Fingerprint64 of "a".."d";
                   to_hash_bucket_fast(["Hello","TensorFlow","2.x"],3) = [0,2,2]) plus the ids they
                   imply for the reference's own test keys "1","2" (tests/keras/test_fm.py:89-97).
* topk_metric / hard_negative / accidental_negative are *procedures* (RandomState(42) inputs +
  a property), restated in tests/test_oracle_golden.py from tests/keras/test_factorized_top_k.py:86-130
  and tests/keras/test_sbcnm.py:16-55; they carry no literal numbers, so nothing to store.

TensorFlow is not importable in the build container, so no vector here was produced by running
the reference; they are the literals its tests assert.
"""
import json, os

HERE = os.path.dirname(os.path.abspath(__file__))

GOLDEN = {
    "cross_kat": {"x0": [[0.1, 0.2, 0.3]], "x": [[0.4, 0.5, 0.6]], "kernel": "ones", "bias": "zeros",
                  "expected": [[0.55, 0.8, 1.05]]},
    "take_long_axis": {"arr": [[0.1, 0.2, 0.3], [0.4, 0.5, 0.6]], "indices": [[0, 1], [2, 1]],
                       "expected": [[0.1, 0.2], [0.6, 0.5]]},
    "exclude": {"scores": [[0.1, 0.2, 0.3], [0.4, 0.5, 0.6]], "identifiers": [[0, 1, 2], [3, 4, 5]],
                "exclude": [[1, 2], [3, 5]], "k": 1, "expected_scores": [[0.1], [0.5]], "expected_ids": [[0], [4]]},
    "farmhash": {
        "fingerprint64": {"a": 12917804110809363939, "b": 11795596070477164822,
                          "c": 11430444447143000872, "d": 4470636696479570465},
        "to_hash_bucket_fast": {"inputs": ["Hello", "TensorFlow", "2.x"], "num_buckets": 3, "expected": [0, 2, 2]},
        "reference_test_keys": [{"key": "1", "num_buckets": 6040, "id": 1529},
                                {"key": "1", "num_buckets": 100, "id": 49},
                                {"key": "2", "num_buckets": 100, "id": 59}],
    },
}

if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(GOLDEN, f, indent=1)
    print("wrote", os.path.join(HERE, "reference_kats.json"))
