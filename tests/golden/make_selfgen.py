"""Regenerates tests/golden/selfgen/*: a seeded synthetic case whose expected distances
come from the ORACLE (oracle/unifrac_oracle.c), not from the reference -- the reference
cannot be built in this image (no Go toolchain).  The files pin the synthetic generator
and the oracle against silent drift (numpy version, refactors).

    python tests/golden/make_selfgen.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from frackyfrac_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "selfgen")
SEED, NS, NL, DENS = 0xF4AC0063, 24, 40, 0.25


def main():
    tree, ptr, idx, val = synth.make(NS, NL, DENS, SEED)
    # give the tree features the reference's own files lack: unequal leaf depths are
    # inherent to a Yule tree; add a multifurcation-free but non-dyadic length and a root length
    tree.branch_len[5] = 0.3
    tree.branch_len[0] = 0.125
    open(os.path.join(OUT, "synth24.tree"), "w").write(tree.newick())
    open(os.path.join(OUT, "synth24.sparse"), "w").write(synth.sparse_text(tree, ptr, idx, val))
    open(os.path.join(OUT, "synth24.dense"), "w").write(synth.dense_text(tree, ptr, idx, val))
    otree = O.parse_newick(open(os.path.join(OUT, "synth24.tree")).read())
    oab = O.parse_sparse_abundance(open(os.path.join(OUT, "synth24.sparse")).read())
    for weighted, name in ((False, "synth24.unweighted.want"), (True, "synth24.weighted.want")):
        d = O.unifrac(oab, otree, weighted)
        assert np.array_equal(d, np.array(O.unifrac_py(oab, otree, weighted)))
        open(os.path.join(OUT, name), "w").write(O.format_output(d))


if __name__ == "__main__":
    main()
