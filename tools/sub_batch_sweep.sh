for r in 1 2; do for sb in 1 2 3 4; do
  v=$(timeout 200 python bench.py --net mobilenet_v1 --sub-batches $sb --steps 50 --no-cpu-baseline --no-steady 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
  echo "subbatches $sb round $r: $v"
done; done
