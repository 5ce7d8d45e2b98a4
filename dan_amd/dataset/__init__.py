"""Input records of the detector trainers — the host side of dataset/dataset_common.py."""
