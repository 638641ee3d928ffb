"""Minimal FASTA reader / writer (the reference uses Biopython's SeqIO, which this image lacks)."""


def read_fasta(path):
    """Yield (identifier, header_rest, sequence).  identifier = first whitespace-delimited token of
    the header line (what SeqIO calls record.id), header_rest = the remainder of that line."""
    ident, rest, chunks = None, "", []
    with open(path, "r") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if ident is not None:
                    yield ident, rest, "".join(chunks)
                head = line[1:].strip()
                parts = head.split(None, 1)
                ident = parts[0] if parts else ""
                rest = parts[1] if len(parts) > 1 else ""
                chunks = []
            elif ident is not None and line:
                chunks.append(line.strip())
    if ident is not None:
        yield ident, rest, "".join(chunks)


def write_record(fh, ident, seq, description="", width=60):
    """One record, sequence wrapped at `width` columns (SeqIO's FASTA writer wraps at 60)."""
    fh.write(">" + ident + ((" " + description) if description else "") + "\n")
    for i in range(0, len(seq), width):
        fh.write(seq[i:i + width] + "\n")
