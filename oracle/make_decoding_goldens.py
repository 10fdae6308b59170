"""tests/golden/decoding.json from the VERBATIM reference decoders (needs /root/reference):
    python -m oracle.make_decoding_goldens"""
import json
import os
import sys

from oracle import decoding_cases as dc
from oracle import reference_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "decoding.json")


def main():
    if not reference_import.available():
        sys.exit("needs /root/reference")
    reference_import.import_reference()
    from virtex.utils.beam_search import AutoRegressiveBeamSearch
    from virtex.utils.nucleus_sampling import AutoRegressiveNucleusSampling
    rec = {}
    for name in dc.CASES:
        tokens, lp = dc.run(AutoRegressiveBeamSearch, AutoRegressiveNucleusSampling, name)
        rec[name] = {"tokens": tokens.tolist(), "logprobs": None if lp is None else lp.tolist()}
        print(name, tuple(tokens.shape))
    with open(OUT, "w") as f:
        json.dump(rec, f)
    print("->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
