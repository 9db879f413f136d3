"""cu2rec_amd -- MI355X-native matrix-factorisation SGD hot path (drop-in for nickgreenquist/cu2rec's
sgd.cu / loss.cu / training.cu path).  The product is libcu2rec_amd.so (C ABI, include/cu2rec_amd.h)
and bin/mf; this package is the Python mirror of the reference's host interface on top of it."""
from ._lib import Config, Cu2recError, Hyper, TrainStats, lib  # noqa: F401
from .api import (SGD_BLOCKSOLVE, SGD_HOGWILD, SGD_ORDERED, SGD_PINGPONG, SGD_SERIAL, DeviceCSR, HostCSR, Model, Ratings, createSparseMatrix,  # noqa: F401
                  default_config, device_count, initialize_normal_array, print_config, read_array, read_config,
                  readCSV, sampler_index, shard_plan, train, write_config, writeCSV, writeToFile)
