"""MI355X-native descriptor matching for the VisualSfM workflow of thuanaislab/visual_slam.

The compute path lives in ``libvsm.so`` (hand-written HIP for gfx950 behind a C ABI,
``include/vsm.h``).  This package is the thin host-side mirror of the reference's Python
interface around that path (``Others/SuperPoint_data/main.py`` helper names and argument
meaning), bound with ctypes.  There is no CPU fallback: without the built library or
without a gfx950 device every compute call raises.
"""
from ._binding import (  # noqa: F401
    VsmError,
    Matcher,
    SlicedDatabase,
    lib,
    lib_path,
    METRIC_DOT,
    METRIC_L2,
    LAYOUT_ROWS,
    LAYOUT_DIM_MAJOR,
    NO_SECOND,
    F32_FAST_EXACT,
    F32_REVERSE_ALL,
    F32_REVERSE_AUTO,
    F32_REVERSE_CANDIDATES,
    F32_MFMA_F32,
    U8_AUTO,
    U8_PACKED,
    U8_TILE_SEPARATE,
    U8_TILE_FUSED,
)
from .host import (  # noqa: F401
    generate_imagePairs,
    generate_exhaustive_pairs,
    sort_scores,
    sort_feature_Or_descriptors,
    get_correct_new_matching_index,
    resort_pair_by_score,
    get_list_matching_pairs,
    save_matching_sfm,
    save_feature_locations,
    read_feature_locations,
    read_sift,
    write_sift,
    write_pmatch,
    get_pmatch,
    get_pmatch_r,
    get_match_count,
    write_fmatch,
    write_ematch,
    get_pose,
    estimate_pose,
    get_imatch,
    get_imatch_r,
    estimate_fundamental,
    nearest_embed,
    mutual_nn_match,
)

__all__ = [
    "VsmError", "Matcher", "SlicedDatabase", "lib", "lib_path",
    "METRIC_DOT", "METRIC_L2", "LAYOUT_ROWS", "LAYOUT_DIM_MAJOR", "NO_SECOND",
    "F32_FAST_EXACT", "F32_MFMA_F32", "F32_REVERSE_AUTO", "F32_REVERSE_CANDIDATES", "F32_REVERSE_ALL", "U8_AUTO", "U8_PACKED", "U8_TILE_SEPARATE", "U8_TILE_FUSED",
    "generate_imagePairs", "generate_exhaustive_pairs", "get_list_matching_pairs",
    "sort_scores", "sort_feature_Or_descriptors", "get_correct_new_matching_index", "resort_pair_by_score",
    "save_matching_sfm", "save_feature_locations", "read_feature_locations",
    "read_sift", "write_sift", "write_pmatch", "get_pmatch", "get_pmatch_r", "get_match_count",
    "write_fmatch", "write_ematch", "get_pose", "estimate_pose", "get_imatch", "get_imatch_r", "estimate_fundamental",
    "nearest_embed", "mutual_nn_match",
]
