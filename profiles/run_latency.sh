#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/run_latency.sh <tag>  -- two PMC passes: how long the scalar cache, LDS, vector
# memory and the instruction fetch keep a wave of k_bounce waiting (SQ_INST_LEVEL_* / SQ_INSTS_* = average latency in cycles)
TAG=${1:-x}
PMC="SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA" bash profiles/run_valu.sh ${TAG}_a
PMC="SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" bash profiles/run_valu.sh ${TAG}_b
PMC="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" bash profiles/run_valu.sh ${TAG}_c
