"""
TEST INFRASTRUCTURE (see oracle/__init__.py): CPU restatement of the reference's reader for Merlin /
Medipix .mib files, src/libertem/io/dataset/mib.py -- header fields, frame layout and the raw ('R64')
pixel decoders, single chip and 2x2 quad.  Written pixel by pixel like the reference's numba loops
(small cases only).  Pinned against tests/golden/mib.npz: frames, SumSigUDF and ApplyMasksUDF results
the REAL reference's MIBDataSet produced from the files of tests/golden/recipes.py:make_mib_case
(tests/golden/generate_golden.py:gen_mib).
"""
import numpy as np


def parse_header(first_bytes, filesize):
    """MIBHeaderReader._parse_header_bytes (mib.py:802-892)"""
    text = first_bytes.decode('ascii', errors='ignore')
    size = int(text.split(',')[2])
    parts = [p for p in text[:size].split(',') if '\x00' not in p]
    mib_dtype = parts[6].lower()
    kind = mib_dtype[0]
    if kind not in 'ur':
        raise ValueError(f"unknown kind: {kind}")
    height, width = int(parts[5]), int(parts[4])
    bits = int(parts[-1])
    if kind == 'u':
        payload = height * width * (int(mib_dtype[1:]) // 8)
    else:
        if bits == 24:
            width //= 2                                   # two 12-bit images (mib.py:829-830)
        payload = int(height * width * {1: 1 / 8, 6: 1, 12: 2, 24: 4}[bits])
    n_chips = int(parts[3])
    lay = parts[7].replace('G', '').split('x')
    layout = (int(lay[0]), int(lay[1]))
    if kind == 'r' and n_chips > 1:
        # raw rows of all chips side by side -> detector shape (mib.py:858-871)
        px = height
        if px * layout[1] * px * layout[0] != height * width:
            raise ValueError(f"invalid sensor layout {layout}")
        height, width = px * layout[1], px * layout[0]
    return dict(header_size_bytes=size, mib_dtype=mib_dtype, mib_kind=kind, bits_per_pixel=bits,
                image_size=(height, width), image_size_bytes=payload,
                sequence_first_image=int(parts[1]), num_images=filesize // (payload + size),
                num_chips=n_chips, sensor_layout=layout)


def declared_dtype(fields):
    """MIBHeaderReader._get_np_dtype (mib.py:771-787): what `dataset.dtype` reports"""
    if fields['mib_kind'] == 'u':
        return np.dtype('>u%d' % (int(fields['mib_dtype'][1:]) // 8))
    return {1: np.dtype('uint64'), 6: np.dtype('uint8'), 12: np.dtype('uint16'),
            24: np.dtype('uint16')}[fields['bits_per_pixel']]


def _decode_row(raw, bits, n_px):
    """decode_r1_swap / decode_r6_swap / decode_r12_swap (mib.py:608-644) for one run of pixels"""
    out = np.zeros(n_px, dtype=np.uint32)
    if bits == 1:
        for stripe in range(len(raw) // 8):
            for byte in range(8):
                b = int(raw[(stripe + 1) * 8 - (byte + 1)])
                for bit in range(8):
                    out[64 * stripe + 8 * byte + bit] = (b >> bit) & 1
    elif bits == 6:
        for i in range(n_px):
            out[(i // 8 + 1) * 8 - i % 8 - 1] = raw[i]
    elif bits == 12:
        for i in range(n_px):
            out[(i // 4 + 1) * 4 - i % 4 - 1] = (int(raw[2 * i]) << 8) + int(raw[2 * i + 1])
    else:
        raise ValueError(bits)
    return out


def decode_frame(payload, fields):
    """One frame's payload bytes -> (H, W) pixel values (uint32)."""
    h, w = fields['image_size']
    bits = fields['bits_per_pixel']
    raw = np.frombuffer(payload, dtype=np.uint8)
    if fields['mib_kind'] == 'u':
        return raw.view(declared_dtype(fields)).reshape(h, w).astype(np.uint32)
    if bits == 24:
        # mib.py:647-665: first image holds the most significant 12 bits
        half = len(raw) // 2
        f12 = dict(fields, bits_per_pixel=12)
        return (decode_frame(raw[:half].tobytes(), f12) << 12) + \
            decode_frame(raw[half:].tobytes(), f12)
    out = np.zeros((h, w), dtype=np.uint32)
    row_bytes = w * {1: 1, 6: 8, 12: 16}[bits] // 8
    if fields['num_chips'] == 4 and fields['sensor_layout'] == (2, 2):
        # mib.py:260-398 + decode_r*_swap_2x2: a raw row = [chip 4 | chip 3 | chip 2 | chip 1],
        # each half a detector row long; chips 3 / 4 fill the bottom half rotated by 180 degrees
        hb = row_bytes // 2
        stride = 4 * hb
        for y in range(h):
            if y < h // 2:
                left = raw[y * stride + 3 * hb: y * stride + 4 * hb]
                right = raw[y * stride + 2 * hb: y * stride + 3 * hb]
                out[y, :w // 2] = _decode_row(left, bits, w // 2)
                out[y, w // 2:] = _decode_row(right, bits, w // 2)
            else:
                yy = h - y - 1
                left = raw[yy * stride + 1 * hb: yy * stride + 2 * hb]
                right = raw[yy * stride: yy * stride + hb]
                out[y, :w // 2] = _decode_row(left, bits, w // 2)[::-1]
                out[y, w // 2:] = _decode_row(right, bits, w // 2)[::-1]
        return out
    for y in range(h):
        out[y] = _decode_row(raw[y * row_bytes:(y + 1) * row_bytes], bits, w)
    return out


def read_files(files, nav_shape, sync_offset=0):
    """{name: bytes} -> (prod(nav), H, W) uint32 frames of the scan: files ordered by the sequence
    number of their first frame (mib.py:1105-1106), `sync_offset` frames skipped (> 0) or blank frames
    inserted at the start (< 0), missing frames at the end blank."""
    parsed = []
    for name, blob in files.items():
        fields = parse_header(blob[:1024], len(blob))
        parsed.append((fields['sequence_first_image'], blob, fields))
    parsed.sort(key=lambda t: t[0])
    frames = []
    for _, blob, fields in parsed:
        hs, ps = fields['header_size_bytes'], fields['image_size_bytes']
        for i in range(fields['num_images']):
            o = i * (hs + ps) + hs
            frames.append(decode_frame(blob[o:o + ps], fields))
    n_nav = int(np.prod(nav_shape))
    out = np.zeros((n_nav,) + frames[0].shape, dtype=np.uint32)
    skip, blank = max(sync_offset, 0), max(-sync_offset, 0)
    src = frames[skip:skip + n_nav - blank]
    for i, f in enumerate(src):
        out[blank + i] = f
    return out, parsed[0][2]
