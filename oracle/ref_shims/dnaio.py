"""TEST INFRASTRUCTURE ONLY -- stand-in for dnaio==1.2.3 (kindel/kindel.py:434,
tests/test_kindel.py:117): a record type and a wrapped-FASTA reader."""


class Sequence:
    def __init__(self, name, sequence, qualities=None):
        self.name, self.sequence, self.qualities = name, sequence, qualities


class _Reader:
    def __init__(self, path):
        self._path = path

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        name, chunks = None, []
        with open(self._path) as fh:
            for line in fh:
                line = line.rstrip("\n")
                if line.startswith(">"):
                    if name is not None:
                        yield Sequence(name, "".join(chunks))
                    name, chunks = line[1:], []
                elif line:
                    chunks.append(line)
        if name is not None:
            yield Sequence(name, "".join(chunks))


def open(path, mode="r"):  # noqa: A001 - mirrors dnaio.open
    return _Reader(str(path))
