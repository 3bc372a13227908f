"""Sentinel-2 spectral response functions (host-side table preparation).

Mirrors reference ``s2_emit/srf.py`` (names, arguments, defaults, exceptions).  The default source
is ESA's xlsx on the web exactly as in the reference (srf.py:6-9); because GPU nodes are usually
offline, a local path is accepted too, and so are ``.csv`` / ``.npz`` exports of the same table.
xlsx files are parsed with pandas when an Excel engine is installed, otherwise with a small
built-in reader (an .xlsx is a zip of XML parts), so no extra dependency is needed.
"""
from __future__ import annotations

import io
import os
import re
import zipfile
from typing import Dict, List, Optional, Tuple
from xml.etree import ElementTree as ET

import numpy as np

DEFAULT_SRF_XLSX_URL = (
    "https://sentiwiki.copernicus.eu/__attachments/1692737/"
    "COPE-GSEG-EOPG-TN-15-0007%20-%20Sentinel-2%20Spectral%20Response%20Functions%202022%20-%203.2.xlsx"
)

S2_BANDS_13 = ["B1", "B2", "B3", "B4", "B5", "B6", "B7", "B8", "B8A", "B9", "B10", "B11", "B12"]

_NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
       "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships",
       "p": "http://schemas.openxmlformats.org/package/2006/relationships"}


class _Workbook:
    """Minimal xlsx reader: sheet names and a sheet as {column header -> list of cell values}."""

    def __init__(self, source):
        if isinstance(source, (bytes, bytearray)):
            source = io.BytesIO(source)
        self._zip = zipfile.ZipFile(source)
        wb = ET.fromstring(self._zip.read("xl/workbook.xml"))
        rels = ET.fromstring(self._zip.read("xl/_rels/workbook.xml.rels"))
        target = {r.get("Id"): r.get("Target") for r in rels.findall("p:Relationship", _NS)}
        self._sheets = {}
        for s in wb.find("m:sheets", _NS).findall("m:sheet", _NS):
            t = target[s.get("{%s}id" % _NS["r"])]
            self._sheets[s.get("name")] = t.lstrip("/") if t.startswith("/") else "xl/" + t
        self.sheet_names = list(self._sheets)
        self._shared = []
        if "xl/sharedStrings.xml" in self._zip.namelist():
            ss = ET.fromstring(self._zip.read("xl/sharedStrings.xml"))
            for si in ss.findall("m:si", _NS):
                self._shared.append("".join(t.text or "" for t in si.iter("{%s}t" % _NS["m"])))

    @staticmethod
    def _col_index(ref: str) -> int:
        n = 0
        for ch in re.match(r"[A-Z]+", ref).group(0):
            n = n * 26 + (ord(ch) - 64)
        return n - 1

    def parse(self, sheet: str) -> Dict[str, list]:
        root = ET.fromstring(self._zip.read(self._sheets[sheet]))
        rows = []
        for row in root.find("m:sheetData", _NS).findall("m:row", _NS):
            cells = {}
            for c in row.findall("m:c", _NS):
                v = c.find("m:v", _NS)
                kind = c.get("t")
                if kind == "inlineStr":
                    val = "".join(t.text or "" for t in c.iter("{%s}t" % _NS["m"]))
                elif v is None or v.text is None:
                    val = None
                elif kind == "s":
                    val = self._shared[int(v.text)]
                elif kind in ("str", "e"):
                    val = v.text
                elif kind == "b":
                    val = bool(int(v.text))
                else:
                    val = float(v.text)
                cells[self._col_index(c.get("r"))] = val
            rows.append(cells)
        if not rows:
            return {}
        header = rows[0]
        table = {}
        for idx, name in header.items():
            if name is None:
                continue
            table[str(name)] = [r.get(idx) for r in rows[1:]]
        return table


def _to_numeric(values) -> np.ndarray:
    """pd.to_numeric(errors='coerce').to_numpy() for a plain list."""
    out = np.full(len(values), np.nan, dtype=float)
    for i, v in enumerate(values):
        if v is None or isinstance(v, bool):
            continue
        try:
            out[i] = float(v)
        except (TypeError, ValueError):
            pass
    return out


def pick_sheet_name(xl, platform: str = "S2A") -> str:
    """First sheet whose name contains 'Spectral Responses' and the platform (srf.py:13-18)."""
    platform = platform.upper()
    candidates = [s for s in xl.sheet_names if ("Spectral Responses" in s and platform in s)]
    if not candidates:
        raise ValueError(f"No sheet containing 'Spectral Responses' and '{platform}' found. Sheets: {xl.sheet_names}")
    return candidates[0]


def _open_table(source, platform):
    """Returns (sheet_name, {column -> numpy array}) from xlsx / csv / npz / dict / DataFrame."""
    if isinstance(source, dict):
        return "<dict>", {k: np.asarray(v) for k, v in source.items()}
    if hasattr(source, "columns") and hasattr(source, "__getitem__"):      # DataFrame
        return "<DataFrame>", {str(c): source[c].to_numpy() for c in source.columns}
    path = os.fspath(source)
    low = path.lower()
    if low.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        return "<npz>", {k: z[k] for k in z.files}
    if low.endswith(".csv") or low.endswith(".txt"):
        import csv
        with open(path, newline="") as f:
            rd = csv.reader(f)
            header = next(rd)
            cols = [[] for _ in header]
            for row in rd:
                for i in range(len(header)):
                    cols[i].append(row[i] if i < len(row) and row[i] != "" else None)
        return "<csv>", {h: c for h, c in zip(header, cols)}
    if re.match(r"^[a-z]+://", path):
        import urllib.request
        with urllib.request.urlopen(path) as resp:      # same network fetch the reference performs
            data = resp.read()
        xl = _Workbook(data)
    else:
        xl = _Workbook(path)
    sheet = pick_sheet_name(xl, platform=platform)
    return sheet, xl.parse(sheet)


def load_s2_srf_from_xlsx(
    xlsx_url: str = DEFAULT_SRF_XLSX_URL,
    platform: str = "S2A",
    bands: Optional[List[str]] = None,
    wavelength_col: str = "SR_WL",
    col_prefix: Optional[str] = None,
) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """Returns dict: band -> (lambda_nm, response) with response > 0 and finite (srf.py:20-52)."""
    bands = bands or S2_BANDS_13
    platform = platform.upper()
    if col_prefix is None:
        col_prefix = f"{platform}_SR_AV_"
    sheet, table = _open_table(xlsx_url, platform)
    if wavelength_col not in table:
        raise KeyError(wavelength_col)
    wavelength_nm = _to_numeric(list(table[wavelength_col]))
    srf_dict: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
    for b in bands:
        col = f"{col_prefix}{b}"
        if col not in table:
            raise KeyError(f"Column '{col}' not found in sheet '{sheet}'.")
        resp = _to_numeric(list(table[col]))
        m = np.isfinite(wavelength_nm) & np.isfinite(resp) & (resp > 0)
        srf_dict[b] = (wavelength_nm[m].astype(float), resp[m].astype(float))
    return srf_dict
