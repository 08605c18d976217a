"""Random line structures for the differential tests of the FASTQ decode kernels: line lengths from empty to several
tiles, CRLF, trailing incomplete entries, 1-4 lines per entry, any sequence line, damaged bytes."""
import numpy as np

ALPH = np.frombuffer(b"ACGTacgt", dtype=np.uint8)


def random_text(rng):
    lpe = int(rng.choice([1, 2, 3, 4, 4, 4, 4]))
    seq_line = int(rng.integers(0, lpe)) if lpe != 4 or rng.random() < 0.2 else 1
    check_plus = bool(lpe >= 3 and rng.random() < 0.7)
    crlf = rng.random() < 0.25
    eol = b"\r\n" if crlf else b"\n"
    style = rng.choice(["short", "mid", "long", "mixed", "tiny", "huge"])
    n_entries = int({"short": rng.integers(1, 4000), "mid": rng.integers(1, 1500), "long": rng.integers(1, 60),
                     "mixed": rng.integers(1, 800), "tiny": rng.integers(1, 9000), "huge": rng.integers(1, 6)}[style])
    def length():
        if style == "short": return int(rng.integers(0, 40))
        if style == "mid": return int(rng.integers(50, 300))
        if style == "long": return int(rng.integers(1000, 40000))
        if style == "tiny": return int(rng.integers(0, 4))
        if style == "huge": return int(rng.integers(20000, 200000))
        return int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 100, 255, 256, 257, 5000, 16384, 16400]))
    parts = []
    for i in range(n_entries):
        ln = length()
        for j in range(lpe):
            if j == seq_line:
                line = rng.choice(ALPH, size=ln).tobytes()
                if j == 0:
                    line = b"@" + line          # (a sequence line that is also the header line keeps the header byte legal...)
            elif j == 0:
                line = b"@" + bytes(rng.integers(33, 127, size=int(rng.integers(0, 30))).astype(np.uint8))
            elif j == 2 and lpe >= 3:
                line = b"+" + (b"" if rng.random() < 0.8 else b"xyz")
            else:
                line = bytes(rng.integers(33, 127, size=ln if rng.random() < 0.9 else int(rng.integers(0, 50))).astype(np.uint8))
            parts.append(line + eol)
    text = b"".join(parts)
    if rng.random() < 0.3:                                     # trailing incomplete entry
        text += b"@part" + (eol if rng.random() < 0.5 else b"") + (b"ACG" if rng.random() < 0.5 else b"")
    buf = np.frombuffer(text, dtype=np.uint8).copy()
    if buf.size and rng.random() < 0.3:                        # damage
        for _ in range(int(rng.integers(1, 4))):
            buf[int(rng.integers(0, buf.size))] = int(rng.choice([ord("N"), ord("x"), ord("@"), ord("+"), 10, 13, 0, 200]))
    return buf, lpe, seq_line, check_plus
