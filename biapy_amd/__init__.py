"""biapy_amd - MI355X (gfx950) native 3D patch U-Net hot path for BiaPy.

Public surface mirrors the reference names for this path:
  biapy_amd.resunet.ResUNet                       <- biapy.models.resunet.ResUNet
  biapy_amd.unet.U_Net                            <- biapy.models.unet.U_Net (2D and 3D)
  biapy_amd.rcan.rcan                             <- biapy.models.rcan.rcan (3D trunk, upscaling_layer=False)
  biapy_amd.tiling.crop_3D_data_with_overlap      <- biapy.data.data_3D_manipulation.crop_3D_data_with_overlap
  biapy_amd.tiling.merge_3D_data_with_overlap     <- biapy.data.data_3D_manipulation.merge_3D_data_with_overlap
  biapy_amd.workflow.SlidingWindowPredictor       <- Base_Workflow.process_test_sample (per-patch branch)
  biapy_amd.chunked.ChunkedPredictor / ChunkGrid  <- chunked_test_pair_data_generator + process_test_sample_by_chunks (in-HBM volumes)
  biapy_amd.train_engine.train_one_epoch/evaluate <- biapy.engine.train_engine.train_one_epoch / evaluate
  biapy_amd.losses / prepost / tta                <- biapy.engine.metrics, biapy.data.norm + semantic_seg Otsu, biapy.data.post_processing TTA
"""
__version__ = "0.1.0"
