"""tce_rvos_amd: MI355X-native TCE-RVOS per-clip forward."""
