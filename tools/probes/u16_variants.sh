for v in "" U16_NO_EPI U16_NO_PRIO ""; do
  if [ -n "$v" ]; then export MTM_LIB_PATH=$PWD/multitemplatematching-python_amd/MTM/variants/libmtm_$v.so; else unset MTM_LIB_PATH; fi
  echo "variant=$v $(python tools/probes/workload.py u16_4k32 2>/dev/null | tail -1 | cut -c100-260)"
done
