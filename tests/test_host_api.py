"""CPU-side tests of the host logic that needs no GPU: option validation and the argument
checks the reference performs before any frame is encoded."""
import pytest


def test_options_builder_validation():  # encode.rs:1418-1455, 1486-1499
    from flac_codec_amd import encode as E

    o = E.Options.default()
    assert (o._c.block_size, o._c.max_partition_order, o._c.max_lpc_order, o._c.mid_side,
            o._c.exhaustive_channel_correlation, o._c.padding) == (4096, 5, 8, 1, 1, 4096)
    f = E.Options.fast()
    assert (f._c.block_size, f._c.max_partition_order, f._c.max_lpc_order, f._c.mid_side,
            f._c.exhaustive_channel_correlation) == (1152, 3, 0, 0, 0)
    b = E.Options.best()
    assert (b._c.block_size, b._c.max_partition_order, b._c.max_lpc_order) == (4096, 6, 12)
    with pytest.raises(E.InvalidBlockSize):
        o.block_size(15)
    with pytest.raises(E.InvalidLpcOrder):
        o.max_lpc_order(33)
    with pytest.raises(E.InvalidLpcOrder):
        o.max_lpc_order(0)
    with pytest.raises(E.InvalidMaxPartitions):
        o.max_partition_order(16)
    with pytest.raises(E.ExcessivePadding):
        o.padding(1 << 24)
    assert o.max_lpc_order(None)._c.max_lpc_order == 0
    assert o.fast_channel_correlation(True)._c.exhaustive_channel_correlation == 0
    assert o.seektable_seconds(0)._c.seektable_mode == 0
    assert o.seektable_frames(7)._c.seektable_value == 7


def test_writer_argument_errors_before_any_gpu_work():  # encode.rs:495, 517-518, 1899-1913
    from flac_codec_amd import encode as E

    o = E.Options.default()
    with pytest.raises(E.InvalidBitsPerSample):
        E.FlacSampleWriter(None, o, 44100, 0, 2)
    with pytest.raises(E.InvalidBitsPerSample):
        E.FlacSampleWriter(None, o, 44100, 33, 2)
    with pytest.raises(E.SamplesNotDivisibleByChannels):
        E.FlacSampleWriter(None, o, 44100, 16, 2, 1001)
    with pytest.raises(E.InvalidTotalSamples):
        E.FlacSampleWriter(None, o, 44100, 16, 2, 0)
    with pytest.raises(E.InvalidSampleRate):
        E.FlacSampleWriter(None, o, 1 << 20, 16, 2)
    with pytest.raises(E.ExcessiveChannels):
        E.FlacSampleWriter(None, o, 44100, 16, 9)
    with pytest.raises(E.ExcessiveTotalSamples):
        E.FlacSampleWriter(None, o, 44100, 16, 1, 1 << 36)
    with pytest.raises(E.InvalidTotalBytes):
        E.FlacByteWriter(None, o, 44100, 16, 2, 0)
    with pytest.raises(E.SamplesNotDivisibleByChannels):
        E.FlacByteWriter(None, o, 44100, 16, 2, 6)


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly, not fall back."""
    import torch

    from flac_codec_amd import encode as E

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.Error):
        E.FlacSampleWriter(None, E.Options.default(), 44100, 16, 2)
