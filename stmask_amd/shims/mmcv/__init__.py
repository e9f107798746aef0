"""Minimal stand-in for the parts of mmcv the reference's eval path touches (datasets/utils.py uses is_str/is_list_of)."""


def is_str(x):
    return isinstance(x, str)


def is_list_of(seq, expected_type):
    return isinstance(seq, list) and all(isinstance(i, expected_type) for i in seq)
