"""Project names and environment-variable names.

Same constants as the reference's voltrix/project/const.py:2-14 (the ``VOLTRIX_*`` flag names are
part of the drop-in surface), plus the gfx950 additions at the bottom.
"""
PROJECT_NAME_FULL = "Voltrix-SpMM"
PROJECT_NAME_ABBR = "Voltrix"
PROJECT_NAME_FULL_LOWER = "voltrix-spmm"
PROJECT_NAME_ABBR_LOWER = "voltrix"

# Environment variables of the reference
DEBUG_FLAG = "VOLTRIX_JIT_DEBUG"
NVCC_COMPILER_FLAG = "VOLTRIX_NVCC_COMPILER"          # honoured as an alias of HIPCC_COMPILER_FLAG
CACHE_DIR_FLAG = "VOLTRIX_CACHE_DIR"
PTXAS_VERBOSE_FLAG = "VOLTRIX_PTXAS_VERBOSE"          # here: adds -Rpass-analysis=kernel-resource-usage
JIT_PRINT_NVCC_COMMAND_FLAG = "VOLTRIX_JIT_PRINT_NVCC_COMMAND"
PRINT_AUTOTUNE_FLAG = "VOLTRIX_PRINT_AUTO_TUNE"

# gfx950 additions -- with the six above, the WHOLE switch surface of the package (INTEGRATION.md lists them with their
# meaning; nothing else is read from the environment, and the native library reads nothing at all)
HIPCC_COMPILER_FLAG = "VOLTRIX_HIPCC_COMPILER"        # path of hipcc (default /opt/rocm/bin/hipcc)
FP32_MODE_FLAG = "VOLTRIX_FP32_MODE"                  # "auto" (default: by the handle) | "fp16" (scaled cast, fp16 MFMA) | "exact" (fp32 MFMA)
PREPROCESS_FLAG = "VOLTRIX_PREPROCESS"                # "fused" (default, GPU) | "fused:sort|bitmap|mixed" | "reference" (CPU + 2 kernels)
TUNE_SPACE_FLAG = "VOLTRIX_TUNE_SPACE"                # "default" | "full" | "none"
TUNED_STORE_FLAG = "VOLTRIX_TUNED_STORE"              # file of persisted tile choices (default <cache dir>/tuned.json)
TUNED_DEFAULTS_FLAG = "VOLTRIX_TUNED_DEFAULTS"        # 0: ignore the shipped bucket defaults (tuned_defaults.json)
HYBRID_FLAG = "VOLTRIX_HYBRID"                        # auto (default) | 1 | 0 | tune : the two-level side-car (hybrid.py)
HYBRID_MIN_SHARE_FLAG = "VOLTRIX_HYBRID_MIN_SHARE"    # fraction of the edges in shared columns the side-car needs
FUSED_FLAG = "VOLTRIX_FUSED"                          # 1: the two-level product as one launch (spmm_fused_kernels.hpp)
CSR_PATH_FLAG = "VOLTRIX_CSR_PATH"                    # auto (default) | 1 | 0 : the CSR row-gather kernel for short-window handles (round 6)
SUPPORTED_FLAGS = (DEBUG_FLAG, NVCC_COMPILER_FLAG, CACHE_DIR_FLAG, PTXAS_VERBOSE_FLAG, JIT_PRINT_NVCC_COMMAND_FLAG,
                   PRINT_AUTOTUNE_FLAG, HIPCC_COMPILER_FLAG, FP32_MODE_FLAG, PREPROCESS_FLAG, TUNE_SPACE_FLAG,
                   TUNED_STORE_FLAG, TUNED_DEFAULTS_FLAG, HYBRID_FLAG, HYBRID_MIN_SHARE_FLAG, FUSED_FLAG, CSR_PATH_FLAG)
