from .scan3r import Scan3RDataset, DeviceBatch  # noqa: F401
