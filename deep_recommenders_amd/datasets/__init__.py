from .movielens import MovieLens, MovielensRanking, TFRecordFile, BytesColumn, parse_int64, parse_bytes  # noqa: F401
