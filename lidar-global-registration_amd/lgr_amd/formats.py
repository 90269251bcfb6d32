"""On-disk formats either side of the hot path, so outputs can be diffed against a run of the
reference and its inputs can be fed to this library:

  * PLY point clouds as `loadPLYFile<PointN>` reads them (reference include/io.h:6-20 -> pcl::PLYReader +
    fromPCLPointCloud2): ascii / binary_little_endian / binary_big_endian, any scalar property types, extra properties
    ignored; x, y, z, normal_x|nx, normal_y|ny, normal_z|nz, intensity|scalar_intensity, curvature are mapped into the
    48-byte PointXYZINormal layout (12 float32: x y z 1 | nx ny nz 0 | intensity curvature 0 0); the list of fields
    found is returned like the reference's `fields` (pointCloudHasNormals looks at it).
  * transformation CSVs (src/common.cpp:83-153: header `reading,gT00..gT33`, row-major 4x4).
  * correspondence CSVs (src/common.cpp:1223-1266).
  * results.csv header / row (src/analysis.cpp:295-328).

Numbers are written like C++ `ostream << float` does by default (6 significant digits, %g).
"""
import os

import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}
_FIELD_SLOT = {"x": 0, "y": 1, "z": 2, "normal_x": 4, "nx": 4, "normal_y": 5, "ny": 5, "normal_z": 6, "nz": 6,
               "intensity": 8, "scalar_intensity": 8, "curvature": 9}
TRANSFORMATION_HEADER = "reading,gT00,gT01,gT02,gT03,gT10,gT11,gT12,gT13,gT20,gT21,gT22,gT23,gT30,gT31,gT32,gT33"
CORRESPONDENCES_HEADER = "query_idx,match_idx,distance,threshold,x_s,y_s,z_s,x_t,y_t,z_t"
RESULTS_HEADER = ("version,descriptor,testname,metric,rmse,correspondences,correct_correspondences,inliers,correct_inliers,"
                  "nr_points,distance_thr,edge_thr,iteration,matching_type,randomness,r_err,t_err,pcd_err,normal_diff,"
                  "corr_uniformity,lrf_type,metric_type,overlap_rmse,alignment_type,keypoint_type,time_cs,time_te,score_type,"
                  "iss_radius_src,iss_radius_tgt,normal_nr_points,reestimate,scale,cluster_k,feature_radius,"
                  "overlap,overlap_area,converged")


def _g(v):
    """C++ default stream formatting of a float / double / int."""
    if isinstance(v, (bool, np.bool_)):
        return "1" if v else "0"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    return "%g" % float(v)


def csv_row(line):
    """Fields of one CSV line as the reference's CSVRow tokenises it (src/csv_parser.cpp:11-25: std::getline, then a plain split
    at every ',' -- no quoting, empty fields kept, a trailing comma yields a last empty field, '\r' stays in the last field).
    Pinned against the reference's own code in tests/test_oracle_ref.py."""
    return line.rstrip("\n").split(",")


# ---------------------------------------------------------------------------------------------------------- PLY
def read_ply(path):
    """-> (points [n x 12 float32], fields [names of the vertex properties that were mapped])."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    cur["props"].append((tok[4], "list", tok[2], tok[3]))
                else:
                    cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
        pts, fields = None, []
        for el in elements:
            has_list = any(p[1] == "list" for p in el["props"])
            if el["name"] != "vertex":
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        f.readline()
                elif not has_list:
                    f.seek(el["count"] * sum(np.dtype(p[1]).itemsize for p in el["props"]), os.SEEK_CUR)
                else:   # variable-length records (faces): walk them
                    end = "<" if fmt == "binary_little_endian" else ">"
                    for _ in range(el["count"]):
                        for p in el["props"]:
                            if p[1] == "list":
                                cnt = int(np.frombuffer(f.read(np.dtype(_PLY_TYPES[p[2]]).itemsize), end + _PLY_TYPES[p[2]])[0])
                                f.seek(cnt * np.dtype(_PLY_TYPES[p[3]]).itemsize, os.SEEK_CUR)
                            else:
                                f.seek(np.dtype(p[1]).itemsize, os.SEEK_CUR)
                continue
            if has_list:
                raise ValueError("%s: list property in the vertex element" % path)
            n = el["count"]
            if fmt == "ascii":
                raw = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2) if n else np.zeros((0, len(el["props"])))
                cols = {name: raw[:, i] for i, (name, _) in enumerate(el["props"])}
            else:
                end = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(name, end + t) for name, t in el["props"]])
                raw = np.frombuffer(f.read(n * dt.itemsize), dt, count=n)
                cols = {name: raw[name] for name, _ in el["props"]}
            pts = np.zeros((n, 12), np.float32)
            pts[:, 3] = 1.0
            for name, _ in el["props"]:
                if name in _FIELD_SLOT:
                    pts[:, _FIELD_SLOT[name]] = cols[name].astype(np.float32)
                    fields.append(name)
        if pts is None:
            raise ValueError("%s: no vertex element" % path)
        return pts, fields


def has_normals(fields):
    """pointCloudHasNormals (include/common.h:465-480, the loader's `normals_available`) with what it actually tests: its
    z flag is set by a normal_x field as well, so normal_x and normal_y decide."""
    s = set(fields)
    return all((a in s) or (b in s) for a, b in (("normal_x", "nx"), ("normal_y", "ny")))


def write_ply(path, pts, binary=True, with_normals=True):
    """PointXYZINormal cloud in PCL's property naming (x y z [normal_x normal_y normal_z] intensity curvature)."""
    pts = np.ascontiguousarray(pts, np.float32)
    names = ["x", "y", "z"] + (["normal_x", "normal_y", "normal_z"] if with_normals else []) + ["intensity", "curvature"]
    cols = [pts[:, _FIELD_SLOT[n]] for n in names]
    with open(path, "wb") as f:
        f.write(("ply\nformat %s 1.0\nelement vertex %d\n" % ("binary_little_endian" if binary else "ascii", pts.shape[0])).encode())
        for n in names:
            f.write(("property float %s\n" % n).encode())
        f.write(b"end_header\n")
        if binary:
            np.stack(cols, 1).astype("<f4").tofile(f)
        else:
            for row in np.stack(cols, 1):
                f.write((" ".join("%.9g" % float(v) for v in row) + "\n").encode())   # 9 digits: reads back to the same float


# ---------------------------------------------------------------------------------------------------------- CSV
def save_transformation(csv_path, name, T):
    """saveTransformation (src/common.cpp:127-153): append one row, header on creation."""
    exists = os.path.exists(csv_path)
    with open(csv_path, "a" if exists else "w") as f:
        if not exists:
            f.write(TRANSFORMATION_HEADER + "\n")
        f.write(name + "".join("," + _g(T[i][j]) for i in range(4) for j in range(4)) + "\n")


def get_transformation(csv_path, name):
    """getTransformation(csv_path, transformation_name) (src/common.cpp:106-125): the first row whose key matches."""
    with open(csv_path) as f:
        for line in f:
            tok = csv_row(line)
            if tok[0] == name:
                return np.array([float(v) for v in tok[1:17]], np.float32).reshape(4, 4)
    raise KeyError(name)


def get_relative_transformation(csv_path, src_filename, tgt_filename):
    """getTransformation(csv_path, src, tgt) (src/common.cpp:83-104): tgt_position^-1 * src_position, or None."""
    pos = {}
    with open(csv_path) as f:
        for line in f:
            tok = csv_row(line)
            if tok[0] in (src_filename, tgt_filename) and len(tok) >= 17:
                try:
                    pos[tok[0]] = np.array([float(v) for v in tok[1:17]], np.float32).reshape(4, 4)   # later rows overwrite
                except ValueError:
                    pass
    if src_filename in pos and tgt_filename in pos:
        return (np.linalg.inv(pos[tgt_filename].astype(np.float64)) @ pos[src_filename].astype(np.float64)).astype(np.float32)
    return None


def save_correspondences(csv_path, src, tgt, corr):
    """saveCorrespondencesToCSV (src/common.cpp:1246-1266); corr: structured array with index_query / index_match / distance / threshold."""
    with open(csv_path, "w") as f:
        f.write(CORRESPONDENCES_HEADER + "\n")
        for c in corr:
            q, m = int(c["index_query"]), int(c["index_match"])
            f.write(",".join([str(q), str(m), _g(c["distance"]), _g(c["threshold"])] + [_g(v) for v in src[q, :3]] + [_g(v) for v in tgt[m, :3]]) + "\n")


def read_correspondences(csv_path):
    """readCorrespondencesFromCSV (src/common.cpp:1223-1244): the first four columns."""
    rows = []
    with open(csv_path) as f:
        f.readline()
        for line in f:
            tok = csv_row(line)
            rows.append((int(tok[0]), int(tok[1]), float(tok[2]), float(tok[3])))
    return np.array(rows, dtype=[("index_query", "<i4"), ("index_match", "<i4"), ("distance", "<f4"), ("threshold", "<f4")])


def results_row(**kw):
    """One row of results.csv in the column order of printAnalysisHeader (src/analysis.cpp:295-328); missing columns
    stay empty (as feature_radius does in the reference when unset)."""
    cols = RESULTS_HEADER.split(",")
    unknown = set(kw) - set(cols)
    if unknown:
        raise KeyError(sorted(unknown))
    return ",".join(_g(kw[c]) if c in kw and not isinstance(kw[c], str) else (kw.get(c, "")) for c in cols)
