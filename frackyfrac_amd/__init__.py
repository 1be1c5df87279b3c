"""frackyfrac_amd -- MI355X-native all-pairs UniFrac (the frcfrc hot path of
fluhus/frackyfrac) behind a C ABI.  See DESIGN.md and include/frackyfrac_amd.h."""
from ._lib import FFError, LIB_PATH, FRCFRC_PATH  # noqa: F401
from .api import (FlatNodes, Plan, Table, Tree, flatten, flatten_device, flatten_leaf_csr, format_float,  # noqa: F401
                  frcfrc_main, iter_pairs, num_pairs, parse_abundance, parse_newick,
                  parse_sparse_abundance, shard_rows, shard_slots, unifrac, unifrac_dists,
                  validate_species, write_distances)

__version__ = "0.1.0"
