"""The extra trunk shapes of the `xshape` build variant (tests/test_gpu_variants.py builds and checks it; __graft_entry__.build() keeps an
existing one up to date).  No imports: readable without pytest / torch."""
XSHAPES = ["6,3,128", "8,2,128", "8,4,256,6,2", "8,4,256,12,4", "8,4,128,5,1", "4,2,256,16,3",
           "6,3,64", "8,4,64,6,2", "8,4,256,10,8", "8,4,128,10,6"]    # round 4: width 64 (8/4 and 4/2 are built in), pos_emb_dir 5..8 (four head k-steps)
