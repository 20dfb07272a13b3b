"""Output / wire formats of the reference's driver, host side only:

* ``save_graphs(samples, save_dir, use_radius=False)`` -- ``src/ConStruct/utils.py:272-315``: one ``graph_<idx>.vtp`` per
  sampled graph.  The reference builds a pyvista ``UnstructuredGrid`` of 2-point cells and saves its extracted surface;
  pyvista / VTK are not needed here: the file is written directly as VTK XML PolyData (ASCII) with the same content -- the
  node coordinates as points, one line cell per ordered pair (i, j) with a non-zero edge class in ``nonzero()`` order (both
  orientations of an undirected edge, as in the reference), cell arrays ``edge_type`` and, with ``use_radius``, ``radius``.
  Graphs without edges are skipped and do not consume an index (:287-289).  Byte-for-byte equality with pyvista's writer
  is not claimed (parity unpinned: pyvista is not in this image); the CPU tests parse the file back.
The Lightning checkpoint key mapping (``model.`` prefix) is ``transformer_model.load_lightning_checkpoint``; it serves
``GraphTransformer`` and the two position-aware classes of ``pos_transformer_model`` alike.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def _fmt(a, fmt):
    return " ".join(fmt % v for v in np.asarray(a).reshape(-1))


def write_vtp(path, points, lines, cell_data):
    """points [n,3] float, lines [m,2] int (point ids), cell_data: name -> [m] array.  VTK XML PolyData, ASCII."""
    points = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    lines = np.asarray(lines, dtype=np.int64).reshape(-1, 2)
    m = lines.shape[0]
    out = ['<?xml version="1.0"?>', '<VTKFile type="PolyData" version="0.1" byte_order="LittleEndian">', "  <PolyData>",
           f'    <Piece NumberOfPoints="{points.shape[0]}" NumberOfVerts="0" NumberOfLines="{m}" NumberOfStrips="0" NumberOfPolys="0">',
           "      <Points>", '        <DataArray type="Float32" NumberOfComponents="3" format="ascii">',
           "          " + _fmt(points, "%.9g"), "        </DataArray>", "      </Points>", "      <Lines>",
           '        <DataArray type="Int64" Name="connectivity" format="ascii">', "          " + _fmt(lines, "%d"), "        </DataArray>",
           '        <DataArray type="Int64" Name="offsets" format="ascii">', "          " + _fmt(2 * (np.arange(m) + 1), "%d"),
           "        </DataArray>", "      </Lines>", "      <CellData>"]
    for name, val in cell_data.items():
        val = np.asarray(val).reshape(-1)
        assert val.shape[0] == m, (name, val.shape, m)
        if np.issubdtype(val.dtype, np.integer):
            out += [f'        <DataArray type="Int64" Name="{name}" format="ascii">', "          " + _fmt(val, "%d"), "        </DataArray>"]
        else:
            out += [f'        <DataArray type="Float32" Name="{name}" format="ascii">', "          " + _fmt(val.astype(np.float32), "%.9g"),
                    "        </DataArray>"]
    out += ["      </CellData>", "    </Piece>", "  </PolyData>", "</VTKFile>", ""]
    with open(path, "w") as f:
        f.write("\n".join(out))


def save_graphs(samples, save_dir, use_radius=False):
    """reference utils.py:272-301.  ``samples``: iterable of collapsed PlaceHolders (``sample_batch`` results; ``E`` integer
    classes, -1 on padding).  Returns the list of files written."""
    os.makedirs(save_dir, exist_ok=True)
    idx, written = 0, []
    for batch in samples:
        for graph in batch.split():
            edges, pos = graph.E.cpu(), graph.pos.cpu()
            ei = torch.nonzero(edges, as_tuple=False).t()
            attr = edges[ei[0], ei[1]]
            if attr.numel() == 0:                      # :287-289: skipped, the index is not consumed
                continue
            cell = {"edge_type": attr.numpy()}
            if use_radius:
                cell["radius"] = graph.radius.cpu()[ei[0], ei[1]].numpy()
            path = os.path.join(save_dir, f"graph_{idx}.vtp")
            write_vtp(path, pos.numpy(), ei.numpy().T, cell)
            written.append(path)
            idx += 1
    return written

