cd $GRAFT_REPO_ROOT
timeout 300 python tools/prof_llm_layer.py --fp8 2>&1 | grep -v amdgpu.ids | grep -E "Name|kernel|aten::|Memcpy|Self CUDA time" | cut -c1-60,120-200 | head -44
