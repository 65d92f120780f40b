"""Byte-level `sensor_msgs/PointCloud2` payload codec (no ROS needed).

The producer side of the boundary is m3d_aggregator: it converts a `pcl::PointCloud<pcl::PointXYZ>`
with `pcl::toPCLPointCloud2` + `pcl_conversions::fromPCL` and publishes it
(/root/reference/m3d/m3d_aggregator/src/m3d_aggregator.cpp:196-209). pcl::PointXYZ is 16 bytes
(x, y, z float32 + 4 bytes padding), so the message carries point_step = 16 with FLOAT32 fields
x@0, y@4, z@8, height = 1 (unorganised), little-endian on every platform the pipeline runs on.
A consumer node decodes with the mirror idiom (m3d_aggregator.cpp:243-246).
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

FLOAT32 = 7  # sensor_msgs/PointField.FLOAT32
FLOAT64 = 8  # sensor_msgs/PointField.FLOAT64


@dataclass
class PointField:
    name: str
    offset: int
    datatype: int = FLOAT32
    count: int = 1


@dataclass
class PointCloud2:
    """The subset of sensor_msgs/PointCloud2 that the registration boundary reads."""
    data: bytes
    width: int
    height: int = 1
    point_step: int = 16
    row_step: int = 0
    is_bigendian: bool = False
    is_dense: bool = True
    frame_id: str = "m3d_test/m3d_link"  # default pointCloudFrame, m3d_aggregator.cpp:152,203
    fields: List[PointField] = field(default_factory=lambda: [PointField("x", 0), PointField("y", 4), PointField("z", 8)])

    @property
    def n(self) -> int:
        return self.width * self.height

    def xyz_offsets(self) -> Tuple[int, int, int]:
        off = {}
        for f in self.fields:
            if f.name in ("x", "y", "z"):
                if f.datatype != FLOAT32 or f.count != 1:
                    raise ValueError(f"field {f.name} must be a single FLOAT32")
                off[f.name] = f.offset
        if len(off) != 3:
            raise ValueError("PointCloud2 lacks x/y/z fields")
        return off["x"], off["y"], off["z"]


def encode_xyz(xyz: np.ndarray, point_step: int = 16, offsets=(0, 4, 8), frame_id="m3d_test/m3d_link",
               big_endian: bool = False) -> PointCloud2:
    """xyz [n,3] float32 -> PointCloud2 with the aggregator's layout (padding bytes zero)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    buf = np.zeros((n, point_step), dtype=np.uint8)
    dt = ">f4" if big_endian else "<f4"
    for a, off in enumerate(offsets):
        buf[:, off:off + 4] = xyz[:, a].astype(dt).view(np.uint8).reshape(n, 4)
    names = ("x", "y", "z")
    return PointCloud2(data=buf.tobytes(), width=n, point_step=point_step, row_step=n * point_step,
                       is_bigendian=big_endian, is_dense=bool(np.isfinite(xyz).all()), frame_id=frame_id,
                       fields=[PointField(names[a], offsets[a]) for a in range(3)])


def decode_xyz(msg: PointCloud2) -> np.ndarray:
    """PointCloud2 -> [n,3] float32 (host-side helper for tests; the library decodes on the device)."""
    ox, oy, oz = msg.xyz_offsets()
    raw = np.frombuffer(msg.data, dtype=np.uint8).reshape(msg.n, msg.point_step)
    dt = ">f4" if msg.is_bigendian else "<f4"
    out = np.empty((msg.n, 3), dtype=np.float32)
    for a, off in enumerate((ox, oy, oz)):
        out[:, a] = np.ascontiguousarray(raw[:, off:off + 4]).view(dt).reshape(-1).astype(np.float32)
    return out


def to_little_endian(msg: PointCloud2) -> PointCloud2:
    """The C ABI takes little-endian payloads (every m3d host is x86/ARM-LE); a big-endian message is
    byte-swapped here on the host before it crosses the boundary."""
    if not msg.is_bigendian:
        return msg
    xyz = decode_xyz(msg)
    out = encode_xyz(xyz, msg.point_step, msg.xyz_offsets(), msg.frame_id, big_endian=False)
    return out


def encode_general(xyz: np.ndarray, fields, point_step: int, width: int = None, height: int = 1, row_pad: int = 0,
                   big_endian: bool = False, frame_id="m3d_test/m3d_link", fill: int = 0xA5) -> PointCloud2:
    """A PointCloud2 with an arbitrary field table (SURVEY §8 row f3 test inputs): `fields` = [PointField] that must
    contain x, y, z as FLOAT32 or FLOAT64 at any offsets; other fields and all padding bytes are filled with `fill`;
    organised clouds (height > 1) get `row_pad` extra bytes at the end of every row."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    width = n if width is None else width
    assert width * height == n
    row_step = width * point_step + row_pad
    buf = np.full((height, row_step), fill, dtype=np.uint8)
    pts = buf[:, :width * point_step].reshape(n, point_step) if row_pad == 0 and height == 1 else None
    col = {"x": 0, "y": 1, "z": 2}
    for f in fields:
        if f.name not in col:
            continue
        dt = (">" if big_endian else "<") + ("f8" if f.datatype == FLOAT64 else "f4")
        sz = 8 if f.datatype == FLOAT64 else 4
        vals = xyz[:, col[f.name]].astype(dt).view(np.uint8).reshape(n, sz)
        for r in range(height):
            rows = buf[r, :width * point_step].reshape(width, point_step)
            rows[:, f.offset:f.offset + sz] = vals[r * width:(r + 1) * width]
    return PointCloud2(data=buf.tobytes(), width=width, height=height, point_step=point_step, row_step=row_step,
                       is_bigendian=big_endian, is_dense=bool(np.isfinite(xyz).all()), frame_id=frame_id, fields=list(fields))
