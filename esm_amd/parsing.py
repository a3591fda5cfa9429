"""FASTA input (reference esm/utils/parsing.py:9-66): `>` headers, sequence lines joined, blank lines and lines
starting with `#` skipped, .gz transparently, ValueError when the input holds no record."""
from __future__ import annotations

import gzip
import io
from pathlib import Path
from typing import Iterator, NamedTuple


class FastaEntry(NamedTuple):
    header: str
    sequence: str


def parse_fasta(fasta_string: str) -> Iterator[FastaEntry]:
    """Records of a FASTA text.  Same results as the reference reader on every input, including its corner cases
    (pinned by tests/golden/text_inputs.json): blank lines and lines starting with `#` vanish wherever they stand,
    header = the text behind `>` stripped, sequence lines are joined as they are, and non-header lines in front of
    the first header end up at the start of the FIRST record's sequence."""
    kept = [ln for ln in fasta_string.splitlines() if ln and ln[0] != "#"]
    heads = [i for i, ln in enumerate(kept) if ln[0] == ">"]
    if not heads:
        raise ValueError("Found no sequences in input")
    ends = heads[1:] + [len(kept)]
    for n, (h, e) in enumerate(zip(heads, ends)):
        body = kept[h + 1:e] if n else kept[:h] + kept[h + 1:e]
        yield FastaEntry(kept[h][1:].strip(), "".join(body))


def read_sequences(path) -> Iterator[FastaEntry]:
    """path: str / Path (.gz allowed) or an open text stream."""
    if isinstance(path, (str, Path)):
        opener = gzip.open if str(path).endswith(".gz") else open
        with opener(path, "rt") as f:
            data = f.read()
    elif isinstance(path, io.TextIOBase) or hasattr(path, "read"):
        data = path.read()
    else:
        raise TypeError(f"cannot read sequences from {type(path).__name__}")
    yield from parse_fasta(data)


def read_first_sequence(path) -> FastaEntry:
    return next(iter(read_sequences(path)))
