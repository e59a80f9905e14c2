cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_embed.py -m gpu -q -k "many_small" 2>&1 | grep -v "^$" | grep "Error\|Mismatch\|Max \|ACTUAL\|DESIRED\|passed\|failed\|assert_allclose" | head -30
