"""Structure-of-arrays repertoire set: what the per-query loop reads through the
reference's ``db_get*`` accessors (/root/reference/src/db.cc:964-997)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

AA = "ACDEFGHIKLMNPQRSTVWY"      # residue codes = index (map_aa, db.cc:33-51,73)
NT = "ACGT"                      # map_nt, db.cc:53-71 (U = T)


@dataclass
class RepertoireSet:
    residues: np.ndarray            # uint8 codes, concatenated
    offsets: np.ndarray             # uint64, n + 1
    v_gene: np.ndarray              # uint32, n (global numbering)
    j_gene: np.ndarray              # uint32, n
    repertoire: np.ndarray          # uint32, n (per-set numbering)
    count: np.ndarray               # uint64, n
    repertoire_ids: List[str] = field(default_factory=list)
    v_names: Optional[List[str]] = None
    j_names: Optional[List[str]] = None
    alphabet: str = AA

    def __post_init__(self):
        self.residues = np.ascontiguousarray(self.residues, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(self.offsets, dtype=np.uint64)
        self.v_gene = np.ascontiguousarray(self.v_gene, dtype=np.uint32)
        self.j_gene = np.ascontiguousarray(self.j_gene, dtype=np.uint32)
        self.repertoire = np.ascontiguousarray(self.repertoire, dtype=np.uint32)
        self.count = np.ascontiguousarray(self.count, dtype=np.uint64)
        n = len(self.repertoire)
        assert len(self.offsets) == n + 1 and len(self.v_gene) == n
        assert len(self.j_gene) == n and len(self.count) == n

    @property
    def n(self) -> int:
        return len(self.repertoire)

    @property
    def n_repertoires(self) -> int:
        return len(self.repertoire_ids)

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets.astype(np.int64))

    @property
    def longest(self) -> int:
        return int(self.lengths.max()) if self.n else 0

    def sequence(self, i: int) -> str:
        a, b = int(self.offsets[i]), int(self.offsets[i + 1])
        return "".join(self.alphabet[c] for c in self.residues[a:b])

    def subset(self, idx) -> "RepertoireSet":
        """Rows `idx` (any numpy index), same repertoire/gene numbering."""
        idx = np.arange(self.n)[idx]
        lens = self.lengths[idx]
        offs = np.zeros(len(idx) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        total = int(offs[-1])
        if total:
            starts = self.offsets[idx].astype(np.int64)
            pos = np.arange(total, dtype=np.int64) - np.repeat(offs[:-1].astype(np.int64), lens)
            res = self.residues[np.repeat(starts, lens) + pos]
        else:
            res = np.zeros(0, dtype=np.uint8)
        return RepertoireSet(res, offs, self.v_gene[idx], self.j_gene[idx],
                             self.repertoire[idx], self.count[idx],
                             list(self.repertoire_ids), self.v_names, self.j_names,
                             self.alphabet)

    def write_tsv_fast(self, path: str, nucleotides: bool = False) -> None:
        """Same columns as write_tsv (minus sequence_id), vectorised with pyarrow:
        10M sequences in seconds.  Used by bench.py to feed the real reference."""
        import pyarrow as pa
        import pyarrow.csv as pc
        col = "junction" if nucleotides else "junction_aa"
        lut = np.frombuffer(self.alphabet.encode(), dtype=np.uint8)
        seq = pa.Array.from_buffers(pa.large_utf8(), self.n,
                                    [None, pa.py_buffer(self.offsets.astype(np.int64)),
                                     pa.py_buffer(lut[self.residues])])

        def names(idx, table):
            return pa.DictionaryArray.from_arrays(pa.array(idx.astype(np.int32)),
                                                  pa.array(table)).dictionary_decode()

        vn = self.v_names or ["V%d" % k for k in range(int(self.v_gene.max(initial=0)) + 1)]
        jn = self.j_names or ["J%d" % k for k in range(int(self.j_gene.max(initial=0)) + 1)]
        tbl = pa.table({"repertoire_id": names(self.repertoire, self.repertoire_ids),
                        "duplicate_count": pa.array(self.count.astype(np.int64)),
                        "v_call": names(self.v_gene, vn), "j_call": names(self.j_gene, jn),
                        col: seq})
        with open(path, "wb") as fh:
            fh.write(("\t".join(tbl.column_names) + "\n").encode())
            pc.write_csv(tbl, fh, write_options=pc.WriteOptions(
                include_header=False, delimiter="\t", quoting_style="none"))

    def write_tsv(self, path: str, nucleotides: bool = False, cdr3: bool = False,
                  crlf: bool = False) -> None:
        """AIRR rearrangement TSV with the columns the path needs."""
        col = ("cdr3" if cdr3 else "junction") + ("" if nucleotides else "_aa")
        eol = "\r\n" if crlf else "\n"
        vn = self.v_names or ["V%d" % k for k in range(int(self.v_gene.max(initial=0)) + 1)]
        jn = self.j_names or ["J%d" % k for k in range(int(self.j_gene.max(initial=0)) + 1)]
        with open(path, "w", newline="") as f:
            f.write("\t".join(["repertoire_id", "sequence_id", "duplicate_count",
                               "v_call", "j_call", col]) + eol)
            for i in range(self.n):
                f.write("\t".join([self.repertoire_ids[int(self.repertoire[i])],
                                   "s%d" % i, str(int(self.count[i])),
                                   vn[int(self.v_gene[i])], jn[int(self.j_gene[i])],
                                   self.sequence(i)]) + eol)
