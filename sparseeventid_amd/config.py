"""Plain-python mirror of the reference's structured configs for the hot path
(/root/reference/src/config/network.py:17-55, framework.py:8-26, data.py:23-75).
Only the values the network construction reads are kept; any object with the same
attribute names (e.g. the reference's hydra config) is accepted by the builders.
"""
from dataclasses import dataclass, field
from enum import Enum


class GrowthRate(Enum):
    multiplicative = 0
    additive = 1


class DownSampling(Enum):
    convolutional = 0
    pooling = 1


class Norm(Enum):
    none = 0
    batch = 1
    layer = 2
    group = 3


class DataMode(Enum):
    dense = 0
    sparse = 1
    graph = 2


@dataclass
class Encoder:
    """src/config/network.py:23-38 defaults."""
    depth: int = 5
    n_initial_filters: int = 32
    n_output_filters: int = 128
    normalization: Norm = Norm.batch
    bias: bool = True
    blocks_per_layer: int = 4
    residual: bool = True
    filter_size: int = 3
    growth_rate: GrowthRate = GrowthRate.additive
    downsampling: DownSampling = DownSampling.convolutional


@dataclass
class Framework:
    mode: DataMode = DataMode.sparse


@dataclass
class Data:
    """dimension 3: dune3d (src/config/data.py), 2: dune2d."""
    dimension: int = 3


@dataclass
class Config:
    encoder: Encoder = field(default_factory=Encoder)
    framework: Framework = field(default_factory=Framework)
    data: Data = field(default_factory=Data)


# Detector geometry of the live loader (src/io/larcv_fetcher.py:23-48) and the north-star grid.
IMAGE_SIZE = {
    "dune3d": [1, 1024, 512, 1280],
    "dune3d_900": [1, 900, 900, 900],
    "dune2d": [3, 1536, 1024],
}

OUTPUT_SHAPE = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}
