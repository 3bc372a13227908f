"""EMIT file loaders.  Mirrors reference ``s2_emit/emit_io.py`` (signatures and return types).

File I/O is outside the accelerated path; these feed K1.  The ENVI
loader is self-contained (header parse + numpy.memmap, BSQ/BIL/BIP -> (H, W, B)) so it works
without the ``spectral`` package; the netCDF reader needs h5py, imported lazily.
"""
from __future__ import annotations

import re
from typing import Optional, Tuple

import numpy as np

_ENVI_DTYPES = {1: np.uint8, 2: np.int16, 3: np.int32, 4: np.float32, 5: np.float64,
                12: np.uint16, 13: np.uint32, 14: np.int64, 15: np.uint64}


def _parse_envi_header(hdr_path: str) -> dict:
    text = open(hdr_path, "r", errors="replace").read()
    fields = {}
    for m in re.finditer(r"^\s*([^=\n]+?)\s*=\s*(\{.*?\}|[^\n]*)", text, flags=re.S | re.M):
        fields[m.group(1).strip().lower()] = m.group(2).strip()
    return fields


def _envi_raw(hdr_path: str, bin_path: str):
    """(memmap of the samples in file order, (lines, samples, bands), interleave)."""
    h = _parse_envi_header(hdr_path)
    lines, samples, bands = int(h["lines"]), int(h["samples"]), int(h["bands"])
    dtype = np.dtype(_ENVI_DTYPES[int(h.get("data type", 4))])
    dtype = dtype.newbyteorder(">" if int(h.get("byte order", 0)) == 1 else "<")
    offset = int(h.get("header offset", 0))
    interleave = h.get("interleave", "bsq").lower()
    raw = np.memmap(bin_path, dtype=dtype, mode="r", offset=offset)[: lines * samples * bands]
    return raw, (lines, samples, bands), interleave


def load_emit_envi_rfl(hdr_path: str, bin_path: str, as_float32: bool = True, device=None):
    """
    Loads EMIT reflectance ENVI pair into memory.
    Returns R: (H, W, B)

    device=None (reference behaviour): a C-contiguous NumPy array; BSQ / BIL files are transposed on the
    host.  device="cuda" (or a torch device): the file goes to the GPU in its native interleave through
    pinned staging chunks and is transposed there to the pixel-major (H, W, B) float32 tensor K1 streams -
    the host never touches the 1.2 GB cube more than once.
    """
    raw, (lines, samples, bands), interleave = _envi_raw(hdr_path, bin_path)
    if device is None:
        if interleave == "bip":
            R = raw.reshape(lines, samples, bands)
        elif interleave == "bil":
            R = raw.reshape(lines, bands, samples).transpose(0, 2, 1)
        else:
            R = raw.reshape(bands, lines, samples).transpose(1, 2, 0)
        R = np.ascontiguousarray(R)
        if as_float32:
            R = R.astype(np.float32, copy=False)
        return R
    from . import _native as nat
    torch = nat.require_gpu()
    dev = torch.device(device)
    if not raw.dtype.isnative:
        raw = raw.astype(raw.dtype.newbyteorder("="))          # rare: big-endian file
    tdtype = {"float32": torch.float32, "float64": torch.float64, "int16": torch.int16, "int32": torch.int32,
              "uint8": torch.uint8, "uint16": torch.uint16, "int64": torch.int64}.get(raw.dtype.name)
    if tdtype is None:
        raise ValueError(f"ENVI data type {raw.dtype} is not supported on the device path")
    flat = torch.empty(raw.shape[0], dtype=tdtype, device=dev)
    chunk = 64 << 20                                              # elements per staging buffer
    stage = [torch.empty(min(chunk, raw.shape[0]), dtype=tdtype).pin_memory() for _ in range(2)]
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    for i, off in enumerate(range(0, raw.shape[0], chunk)):
        n = min(chunk, raw.shape[0] - off)
        buf = stage[i % 2]
        evs[i % 2].synchronize()                                 # previous copy out of this buffer finished
        buf[:n].numpy()[...] = raw[off:off + n]                  # disk/page cache -> pinned
        flat[off:off + n].copy_(buf[:n], non_blocking=True)
        evs[i % 2].record()
    if interleave == "bip":
        R = flat.reshape(lines, samples, bands)
    elif interleave == "bil":
        R = flat.reshape(lines, bands, samples).permute(0, 2, 1)
    else:
        R = flat.reshape(bands, lines, samples).permute(1, 2, 0)
    if as_float32 and R.dtype != torch.float32:
        R = R.to(torch.float32)
    return R.contiguous()


def load_emit_wavelengths_from_nc(
    nc_path: str,
    wavelengths_key: str = "sensor_band_parameters/wavelengths",
    good_key: str = "sensor_band_parameters/good_wavelengths",
) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """
    Returns (emit_wavelengths_nm, good_mask_bool_or_None)

    EMIT L2A netCDF files are HDF5 containers: the band centres (float32, nm) and the per-band
    ``good_wavelengths`` flag live in the ``sensor_band_parameters`` group (reference emit_io.py:18-31).
    """
    import h5py                                    # lazy: only this loader needs it
    with h5py.File(nc_path, "r") as nc:
        wavelengths = np.asarray(nc[wavelengths_key][:], dtype=np.float32)
        flags = nc[good_key][:] if good_key in nc else None
    return wavelengths, (None if flags is None else np.asarray(flags).astype(bool))
