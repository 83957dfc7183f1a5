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

# gfx950 additions
HIPCC_COMPILER_FLAG = "VOLTRIX_HIPCC_COMPILER"        # path of hipcc (default /opt/rocm/bin/hipcc)
OFFLOAD_ARCH_FLAG = "VOLTRIX_OFFLOAD_ARCH"            # default gfx950
FP32_MODE_FLAG = "VOLTRIX_FP32_MODE"                  # "fp16" (default: cast, fp16 MFMA) | "exact" (fp32 MFMA)
PREPROCESS_FLAG = "VOLTRIX_PREPROCESS"                # "fused" (default, GPU) | "reference" (CPU + 2 kernels)
CSR_PATH_FLAG = "VOLTRIX_CSR_PATH"                    # fused preprocess rank algorithm: unset = auto | "sort" | "bitmap" | "mixed" (read in libvoltrix_hip.so)
TUNE_SPACE_FLAG = "VOLTRIX_TUNE_SPACE"                # "default" | "full" | "none"
DISABLE_JIT_FLAG = "VOLTRIX_DISABLE_JIT"              # 1: use the ahead-of-time libvoltrix_hip.so only
