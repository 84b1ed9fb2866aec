"""Scene files either side of the operator: the reference's parquet layout and the INRIA
3D-Gaussian-splatting PLY, to and from the two tensors the operator consumes:
    point_cloud (N,3) f32   point_cloud_features (N,56) f32 = [q xyzw | log-scale | opacity logit | R,G,B SH x16]

* parquet: one row per valid point, columns x,y,z, cov_q0-3, cov_s0-2, alpha0, r_sh0-15, g_sh0-15, b_sh0-15
  (reference GaussianPointCloudScene.py:132-146, 182-210); needs pandas + pyarrow.
* INRIA PLY (binary little endian, all float32): x,y,z,nx,ny,nz,f_dc_0-2,f_rest_0-44,opacity,scale_0-2,rot_0-3 with the
  rotation stored w,x,y,z and f_rest channel-major (reference GaussianPointCloudScene.py:148-180 writes it,
  benchmark/inference_benchmark.py:21-81 reads it).  Read and written with numpy only (no plyfile).
* `preallocate` reproduces max_num_points_ratio (GaussianPointCloudScene.py:28-37): extra rows marked invalid so
  that densification can write into them.
"""
from typing import Optional, Tuple

import numpy as np

FEATURE_COLUMNS = ([f"cov_q{i}" for i in range(4)] + [f"cov_s{i}" for i in range(3)] + ["alpha0"] +
                   [f"r_sh{i}" for i in range(16)] + [f"g_sh{i}" for i in range(16)] + [f"b_sh{i}" for i in range(16)])
PLY_PROPERTIES = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] +
                  ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def save_parquet(path: str, point_cloud: np.ndarray, point_cloud_features: np.ndarray,
                 point_invalid_mask: Optional[np.ndarray] = None) -> None:
    import pandas as pd
    pc = np.asarray(point_cloud, dtype=np.float32)
    ft = np.asarray(point_cloud_features, dtype=np.float32)
    if point_invalid_mask is not None:
        keep = np.asarray(point_invalid_mask) == 0
        pc, ft = pc[keep], ft[keep]
    df = pd.concat([pd.DataFrame(pc, columns=["x", "y", "z"]), pd.DataFrame(ft, columns=FEATURE_COLUMNS)], axis=1)
    df.to_parquet(path)


def load_parquet(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (point_cloud (N,3), point_cloud_features (N,56)); raises if the file has no feature columns
    (a bare x,y,z[,r,g,b] initialisation cloud needs the reference's KD-tree initialiser, which is out of scope)."""
    import pandas as pd
    df = pd.read_parquet(path)
    if not set(FEATURE_COLUMNS).issubset(df.columns):
        raise ValueError(f"{path} holds no trained features (columns {FEATURE_COLUMNS[0]}..{FEATURE_COLUMNS[-1]} missing)")
    return (np.ascontiguousarray(df[["x", "y", "z"]].to_numpy(dtype=np.float32)),
            np.ascontiguousarray(df[FEATURE_COLUMNS].to_numpy(dtype=np.float32)))


def features_to_ply_columns(point_cloud: np.ndarray, features: np.ndarray) -> np.ndarray:
    n = point_cloud.shape[0]
    sh = features[:, 8:].reshape(n, 3, 16)
    cols = [point_cloud, np.zeros((n, 3), np.float32), sh[..., 0], sh[..., 1:].reshape(n, 45),
            features[:, 7:8], features[:, 4:7], features[:, [3, 0, 1, 2]]]          # rotation stored w,x,y,z
    return np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")


def save_inria_ply(path: str, point_cloud: np.ndarray, point_cloud_features: np.ndarray,
                   point_invalid_mask: Optional[np.ndarray] = None) -> None:
    pc = np.asarray(point_cloud, dtype=np.float32)
    ft = np.asarray(point_cloud_features, dtype=np.float32)
    if point_invalid_mask is not None:
        keep = np.asarray(point_invalid_mask) == 0
        pc, ft = pc[keep], ft[keep]
    data = features_to_ply_columns(pc, ft)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {pc.shape[0]}"] + \
             [f"property float {name}" for name in PLY_PROPERTIES] + ["end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(data.tobytes())


def load_inria_ply(path: str, normalise_rotation: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """INRIA point_cloud.ply -> (point_cloud, point_cloud_features), as benchmark/inference_benchmark.py:21-81."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path} is not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("PLY header is not terminated")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"vertex property {tok[-1]} is {tok[1]}, expected float")
                props.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("only binary_little_endian PLY with a vertex element is supported")
        raw = np.frombuffer(fh.read(4 * n * len(props)), dtype="<f4").reshape(n, len(props))
    col = {name: i for i, name in enumerate(props)}
    rest = sorted([p for p in props if p.startswith("f_rest_")], key=lambda s: int(s.split("_")[-1]))
    if len(rest) != 45:
        raise ValueError(f"expected 45 f_rest_* properties (SH degree 3), found {len(rest)}")
    xyz = raw[:, [col["x"], col["y"], col["z"]]]
    rot = raw[:, [col[f"rot_{i}"] for i in range(4)]]
    rot = np.roll(rot, -1, axis=1)                                         # w,x,y,z -> x,y,z,w
    if normalise_rotation:
        rot = rot / np.linalg.norm(rot, axis=1, keepdims=True)
    scale = raw[:, [col[f"scale_{i}"] for i in range(3)]]
    opacity = raw[:, [col["opacity"]]]
    extra = raw[:, [col[p] for p in rest]].reshape(n, 3, 15)
    sh = [np.concatenate([raw[:, [col[f"f_dc_{c}"]]], extra[:, c, :]], axis=1) for c in range(3)]
    feats = np.concatenate([rot, scale, opacity] + sh, axis=1).astype(np.float32)
    return np.ascontiguousarray(xyz, dtype=np.float32), np.ascontiguousarray(feats)


def preallocate(point_cloud: np.ndarray, features: np.ndarray, max_num_points_ratio: Optional[float]):
    """Rows for densification: returns (point_cloud, features, point_invalid_mask i8, point_object_id i32) with
    int(N * ratio) rows of which the first N are valid (GaussianPointCloudScene.py:28-37)."""
    n = point_cloud.shape[0]
    total = n if not max_num_points_ratio else max(n, int(n * max_num_points_ratio))
    pc = np.zeros((total, 3), np.float32)
    ft = np.zeros((total, 56), np.float32)
    pc[:n], ft[:n] = point_cloud, features
    mask = np.ones(total, np.int8)
    mask[:n] = 0
    return pc, ft, mask, np.zeros(total, np.int32)


def merge_scenes(scenes):
    """Several loaded scenes as ONE point set with one pose row per scene, the way the reference's visualiser feeds the
    operator (visualizer.py:292-323): rows concatenated in order, point_object_id = index of the source scene.
    `scenes` is a list of (point_cloud, point_cloud_features) or (point_cloud, features, point_invalid_mask).
    Returns (point_cloud, features, point_invalid_mask i8, point_object_id i32)."""
    pcs, fts, masks, objs = [], [], [], []
    for k, sc in enumerate(scenes):
        pc, ft = np.asarray(sc[0], np.float32), np.asarray(sc[1], np.float32)
        mask = np.zeros(pc.shape[0], np.int8) if len(sc) < 3 or sc[2] is None else np.asarray(sc[2], np.int8)
        pcs.append(pc); fts.append(ft); masks.append(mask); objs.append(np.full(pc.shape[0], k, np.int32))
    return (np.ascontiguousarray(np.concatenate(pcs)), np.ascontiguousarray(np.concatenate(fts)),
            np.ascontiguousarray(np.concatenate(masks)), np.ascontiguousarray(np.concatenate(objs)))
