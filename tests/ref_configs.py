"""Configuration VALUES of the reference's shipped YAML files and of the Spark-TTS BiCodec checkpoint, restated as Python data so that
the `-m gpu` tests can write the files the reference's constructors read (`/root/reference` does not exist on the GPU box).

    HCODEC15_CONFIG   QuarkAudio-HCodec/HCodec-1.5/conf/config_adaptive_v3.yaml
    HCODEC20_CONFIG   QuarkAudio-HCodec/HCodec-2.0/conf/large_12.5hz_config.yaml
tests/test_file_boundary_cpu.py asserts (in the build container, where the reference is mounted) that both equal
`yaml.safe_load` of the shipped file, key for key.

    BICODEC_CONFIG    `BiCodec/config.yaml` of the SparkAudio/Spark-TTS-0.5B download the reference points `codec_ckpt_dir` at
                      (QuarkAudio-UniSE/conf/config.yaml:120; bicodec.py:80).  The file is NOT in the reference tree: these are the
                      published values [upstream-memory]; the same test builds the reference's own modules from these blocks
                      (`Encoder(**config["encoder"])` ... as bicodec.py:82-87 does) to show every block is accepted by their constructors.
"""

HCODEC15_CONFIG = {
    "ckpt_path": "./checkpoints/hcode_1.5_adaptive_4+4.pt",
    "encoder_config": {
        "ratios": [2, 4, 5, 8],
        "encoder": dict(causal=False, n_residual_layers=1, norm="weight_norm", pad_mode="reflect", lstm=6, dimension=512, channels=1,
                        n_filters=32, ratios=[2, 4, 5, 8], activation="ELU", kernel_size=7, residual_kernel_size=3, last_kernel_size=7,
                        dilation_base=2, true_skip=False, compress=2, use_transformer=True),
        "semantic_encoder": dict(input_channels=1024, encode_channels=1024, out_channels=512, channel_ratios=[1, 1], strides=[2, 1]),
    },
    "decoder_config": {
        "decoder": dict(input_channels=1024, dim=1024, intermediate_dim=2304),
        "semantic_decoder": dict(code_dim=512, output_channels=1024, decode_channels=1024, channel_ratios=[1, 1], strides=[2, 1]),
    },
    "quantizer_config": {
        "quantizer": dict(dim=512, codebook_size=1024, num_quantizers=4, decay=0.99, kmeans_init=True, kmeans_iters=50, quantize_dropout=True),
        "semantic_quantizer": dict(dim=512, codebook_size=1024, num_quantizers=4, decay=0.99, kmeans_init=True, kmeans_iters=50,
                                   quantize_dropout=True),
    },
    "adaptive_config": {
        "training": False, "use_similarity_alignment": True, "use_dynamic_similarity_threshold": False, "infer_using_dynamic_threshold": False,
        "similarity_threshold": 0.7, "similarity_threshold_lower": 0.7, "similarity_threshold_upper": 1.0, "max_tokens_per_group": 8,
        "manual_threshold": 0.6, "use_query_token_aggregator": True,
        "aggregators": {
            "semantic_aggregator": dict(dim=512, in_out_dim=512, num_heads=8, num_layers=32, dim_feedforward=2048, causal=False,
                                        use_mean_pooling_init=True, context_frames=16),
            "acoustic_aggregator": dict(dim=512, in_out_dim=512, num_heads=8, num_layers=32, dim_feedforward=2048, causal=False,
                                        use_mean_pooling_init=True, context_frames=16),
        },
        "use_bottleneck_transformer": True,
        "transformer_kwargs": dict(d_model=1024, num_heads=8, num_layers=32, causal=False, layer_scale=0.01, context=16, conv_layout=True,
                                   max_period=10000, gating="none", norm="layer_norm", positional_embedding="rope", dim_feedforward=2048,
                                   input_dimension=1024, output_dimensions=[1024]),
    },
}

HCODEC20_CONFIG = {
    "sampling_rate": 48000,
    "encoder_config": dict(dim=1536, intermediate_dim=4608, dimension=512, n_fft=1920, hop_length=960, convnext_layers=24, transformer_layers=2,
                           target_frame_rate=12.5, causal=False),
    "decoder_config": dict(input_channels=1024, dim=1536, intermediate_dim=4608, convnext_layers=32, transformer_layers=2, n_fft=1920,
                           hop_length=960, target_frame_rate=12.5, causal=False),
    "quantizer_config": dict(dim=512, codebook_size=1024, num_quantizers=16, decay=0.99, kmeans_init=True, kmeans_iters=50, quantize_dropout=False),
    "semantic_encoder_config": dict(input_channels=768, encode_channels=1536, out_channels=512, channel_ratios=[1, 1, 1], strides=[2, 1, 2]),
    "semantic_decoder_config": dict(code_dim=512, output_channels=768, decode_channels=1536, channel_ratios=[1, 1, 1], strides=[2, 1, 2]),
}

BICODEC_CONFIG = {
    "audio_tokenizer": {
        "mel_params": dict(sample_rate=16000, n_fft=1024, win_length=640, hop_length=320, mel_fmin=10, mel_fmax=None, num_mels=128),
        "encoder": dict(input_channels=1024, vocos_dim=384, vocos_intermediate_dim=2048, vocos_num_layers=12, out_channels=1024,
                        sample_ratios=[1, 1]),
        "decoder": dict(input_channel=1024, channels=1536, rates=[8, 5, 4, 2], kernel_sizes=[16, 11, 8, 4]),
        "quantizer": dict(input_dim=1024, codebook_size=8192, codebook_dim=8, commitment=0.25, codebook_loss_weight=2.0, use_l2_normlize=True,
                          threshold_ema_dead_code=0.2),
        "speaker_encoder": dict(input_dim=128, out_dim=1024, latent_dim=128, token_num=32, fsq_levels=[4, 4, 4, 4, 4, 4], fsq_num_quantizers=1),
        "prenet": dict(input_channels=1024, vocos_dim=384, vocos_intermediate_dim=2048, vocos_num_layers=12, out_channels=1024,
                       condition_dim=1024, sample_ratios=[1, 1], use_tanh_at_final=False),
        "postnet": dict(input_channels=1024, vocos_dim=384, vocos_intermediate_dim=2048, vocos_num_layers=6, out_channels=1024,
                        use_tanh_at_final=False),
    },
}
# top level of the Spark-TTS config.yaml that BiCodecTokenizer.get_ref_clip reads (audio_tokenizer.py:62-66) [upstream-memory]
SPARKTTS_CONFIG = dict(sample_rate=16000, ref_segment_duration=6, latent_hop_length=320)


def small_bicodec_config():
    """The same blocks at a size a test can run in seconds (the reference's constructors accept any widths that agree)."""
    import copy

    c = copy.deepcopy(BICODEC_CONFIG)
    a = c["audio_tokenizer"]
    a["quantizer"].update(input_dim=64, codebook_size=128)
    a["speaker_encoder"].update(out_dim=64, latent_dim=32)
    a["prenet"].update(input_channels=64, out_channels=64, condition_dim=64, vocos_dim=32, vocos_intermediate_dim=64, vocos_num_layers=2)
    a["decoder"].update(input_channel=64, channels=512)
    return c
