# A/B of library variants inside one gpurun call: bash profiles/ab_lib.sh "<bench args>" variantA variantB ...
# ("prod" = the product library; others are lib/libflmr_hip_<name>.so from profiles/build_variant.py)
ARGS=$1; shift
L=retrieval-augmented-visual-question-answering_amd/lib
cp $L/libflmr_hip.so /tmp/libflmr_hip.keep
for round in $(seq ${ROUNDS:-2}); do
for v in "$@"; do
  if [ "$v" = prod ]; then cp /tmp/libflmr_hip.keep $L/libflmr_hip.so; else cp $L/libflmr_hip_$v.so $L/libflmr_hip.so; fi
  echo -n "[$v] "; bash profiles/ab_one.sh "$ARGS"
done
done
cp /tmp/libflmr_hip.keep $L/libflmr_hip.so
