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
    header, seq, found = None, [], False
    for line in fasta_string.splitlines():
        if not line or line[0] == "#":
            continue
        if line.startswith(">"):
            if header is not None:
                found = True
                yield FastaEntry(header, "".join(seq))
                seq = []
            header = line[1:].strip()
        else:
            seq.append(line)
    if header is not None:
        found = True
        yield FastaEntry(header, "".join(seq))
    if not found:
        raise ValueError("Found no sequences in input")


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
