"""Binary STL ingest: stands in for the github.com/hschendel/stl v1.0.4
dependency that go.mod:6 lists but no file of the reference imports.  Layout: 80-byte header,
uint32 triangle count, then 50 bytes per triangle {normal 3xf32, 3 vertices 3xf32, uint16}."""
import numpy as np

_REC = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
assert _REC.itemsize == 50


def load_binary_stl(path):
    """-> (N, 3, 3) float64 vertices, facet order preserved."""
    with open(path, "rb") as f:
        head = f.read(84)
        if len(head) < 84:
            raise ValueError(f"{path}: truncated STL header")
        n = int(np.frombuffer(head[80:84], dtype="<u4")[0])
        body = f.read(n * 50)
    if len(body) != n * 50:
        raise ValueError(f"{path}: expected {n} facets ({n * 50} bytes), file holds {len(body)} bytes")
    rec = np.frombuffer(body, dtype=_REC, count=n)
    return np.ascontiguousarray(rec["v"].astype(np.float64))


def save_binary_stl(path, vertices, header=b"raytracer-go_amd"):
    v = np.asarray(vertices, dtype=np.float64)
    if v.ndim != 3 or v.shape[1:] != (3, 3):
        raise ValueError("vertices must have shape (N, 3, 3)")
    rec = np.zeros(v.shape[0], dtype=_REC)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    rec["n"] = np.divide(n, ln, out=np.zeros_like(n), where=ln > 0)
    rec["v"] = v
    with open(path, "wb") as f:
        f.write(header[:80].ljust(80, b"\0"))
        f.write(np.uint32(v.shape[0]).tobytes())
        f.write(rec.tobytes())
