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
    pinned staging chunks and is transposed there (hsr_interleave_to_bip) to the pixel-major (H, W, B) tensor K1
    streams - the host never touches the 1.2 GB cube more than once.
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
    f = EnviCubeFile(hdr_path, bin_path)
    R = f.to_device(device)
    from . import _native as nat
    torch = nat.require_gpu()
    if as_float32 and R.dtype != torch.float32:
        R = R.to(torch.float32)
    return R


_TORCH_NAMES = ("float32", "float64", "int16", "int32", "uint8", "uint16", "int64")


class EnviCubeFile:
    """An ENVI cube on disk, ready to be fed to the GPU in its native interleave.

    The host touches the samples once - file (page cache) -> pinned staging buffer, no transpose, no dtype change -;
    the copy engine moves them; ``hsr_interleave_to_bip`` turns BIL / BSQ into the pixel-major (H, W, B) tensor K1
    streams.  float32 files give a float32 cube, uint16 files a uint16 cube (the reference's tile format, decoded
    inside K1), int16 files a float32 cube (converted in the transpose).  ``SpectralFusion.stream()`` takes these
    objects in place of cube arrays; ``load_emit_envi_rfl(..., device=...)`` is ``EnviCubeFile(...).to_device()``."""

    def __init__(self, hdr_path: str, bin_path: str):
        raw, (self.lines, self.samples, self.bands), self.interleave = _envi_raw(hdr_path, bin_path)
        if not raw.dtype.isnative:
            raw = raw.astype(raw.dtype.newbyteorder("="))          # rare: big-endian file
        if raw.dtype.name not in _TORCH_NAMES:
            raise ValueError(f"ENVI data type {raw.dtype} is not supported on the device path")
        if self.interleave not in ("bip", "bil", "bsq"):
            raise ValueError(f"unknown ENVI interleave {self.interleave!r}")
        self.raw = raw

    @property
    def shape(self):
        return (self.lines, self.samples, self.bands)

    def _torch_dtype(self, torch):
        return getattr(torch, self.raw.dtype.name)

    def cube_dtype(self, torch):
        """dtype of the pixel-major device cube this file becomes."""
        name = self.raw.dtype.name
        return torch.uint16 if name == "uint16" else (torch.float32 if name in ("float32", "int16") else self._torch_dtype(torch))

    def new_staging(self, torch):
        """A pinned host buffer that holds the whole file in file order."""
        return torch.empty(self.raw.shape[0], dtype=self._torch_dtype(torch), pin_memory=True)

    def stage(self, pinned, threads: int = 8):
        """file -> pinned buffer, file order (the one host pass over the samples).  The copy is a plain memcpy out of
        the page cache; one thread moves ~28 GB/s on the GPU boxes' hosts - half the PCIe rate - so it is cut into
        ``threads`` slices copied concurrently (NumPy releases the GIL inside the copy)."""
        dst = pinned.numpy()
        n = self.raw.shape[0]
        if threads <= 1 or n < (1 << 22):
            dst[...] = self.raw
            return pinned
        from concurrent.futures import ThreadPoolExecutor
        step = -(-n // threads)
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(lambda o: np.copyto(dst[o:o + step], self.raw[o:o + step]), range(0, n, step)))
        return pinned

    def to_bip(self, raw_dev, out=None):
        """Device tensor in file order -> pixel-major (lines, samples, bands) tensor (stream ordered)."""
        from . import _native as nat
        from . import _engine as eng
        torch = nat.require_gpu()
        lines, samples, bands = self.shape
        odt = self.cube_dtype(torch)
        if self.interleave == "bip":
            R = raw_dev.reshape(lines, samples, bands)
            return R if R.dtype == odt else R.to(odt)
        code = {"float32": 0, "uint16": 2, "int16": 3}.get(self.raw.dtype.name)
        if code is None:                         # exotic sample types: torch's generic permute
            perm = (0, 2, 1) if self.interleave == "bil" else (1, 2, 0)
            shp = (lines, bands, samples) if self.interleave == "bil" else (bands, lines, samples)
            return raw_dev.reshape(shp).permute(perm).contiguous()
        if out is None:
            out = torch.empty((lines, samples, bands), dtype=odt, device=raw_dev.device)
        with eng._launch(raw_dev) as st:
            nat.check(nat.load().hsr_interleave_to_bip(raw_dev.data_ptr(), code, 1 if self.interleave == "bil" else 2, lines,
                                                       samples, bands, out.data_ptr(), 0 if odt == torch.float32 else 2, st),
                      "hsr_interleave_to_bip")
        return out

    def to_device(self, device="cuda"):
        """The whole file -> (H, W, B) device tensor through two pinned staging chunks."""
        from . import _native as nat
        torch = nat.require_gpu()
        dev = torch.device(device)
        raw = self.raw
        flat = torch.empty(raw.shape[0], dtype=self._torch_dtype(torch), device=dev)
        chunk = 64 << 20                                              # elements per staging buffer
        stage = [torch.empty(min(chunk, raw.shape[0]), dtype=flat.dtype).pin_memory() for _ in range(2)]
        evs = [torch.cuda.Event(), torch.cuda.Event()]
        with torch.cuda.device(dev):
            for i, off in enumerate(range(0, raw.shape[0], chunk)):
                n = min(chunk, raw.shape[0] - off)
                buf = stage[i % 2]
                evs[i % 2].synchronize()                                 # previous copy out of this buffer finished
                buf[:n].numpy()[...] = raw[off:off + n]                  # disk/page cache -> pinned
                flat[off:off + n].copy_(buf[:n], non_blocking=True)
                evs[i % 2].record()
            return self.to_bip(flat).contiguous()


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
