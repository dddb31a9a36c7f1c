"""Build a variant library of the CURRENT sources with extra compiler flags, for same-box A/B runs via $SPARF_LIB
(tools/ab_kernels.sh, tools/evidence.sh ab / dgradprobes):

    python tools/build_flag_variant.py <tag> "<flags>" <translation units the flags affect, comma-separated>
    python tools/build_flag_variant.py nodefer "-DSP_BWD_DEFER=0" mlp_bwd.hip,mlp_bwd_q8.hip
    python tools/build_flag_variant.py p0 "-DSP_PROF" mlp_bwd.hip            # wave-time accounting of the data-gradient kernel

-> sparf_amd/libsparf_hip_<tag>.so (objects in sparf_amd/csrc/build_<tag>/; the other units are linked from the default build).
(tools/build_variant.py builds another git REVISION instead.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparf_amd import build as B                                     # noqa: E402

if __name__ == "__main__":
    tag, flags, only = sys.argv[1], sys.argv[2].split(), sys.argv[3].split(",")
    print(B.build(tag="_" + tag, out=os.path.join(B.HERE, f"libsparf_hip_{tag}.so"), extra_flags=flags, only=only, verbose=False))
