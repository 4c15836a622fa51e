//! Golden .flac files made by the REFERENCE encoder itself (tuffy/flac-codec 1.3.2), for the day a Rust toolchain is at hand:
//!   python3 tests/golden/make_golden.py --export-inputs tests/golden/ref_inputs     (raw PCM + manifest.tsv)
//!   cargo run --release --manifest-path tools/ref_golden/Cargo.toml -- tests/golden/ref_inputs tests/golden/ref_flac
//! tests/test_ref_golden.py then compares the oracle's and the GPU path's streams with these files byte for byte.
//! manifest.tsv: name <TAB> preset (default|fast|best) <TAB> max_lpc_order (-1: the preset's, 0: none) <TAB> padding (-1: none,
//! 0: the preset's) <TAB> sample rate <TAB> bits per sample <TAB> channels;  <name>.raw holds the interleaved samples as
//! little-endian ceil(bps / 8)-byte integers -- what FlacByteWriter<_, LittleEndian>::write takes (encode.rs:359).
use flac_codec::byteorder::LittleEndian;
use flac_codec::encode::{FlacByteWriter, Options};
use std::{env, error::Error, fs, io::Write, path::Path};

fn main() -> Result<(), Box<dyn Error>> {
    let args: Vec<String> = env::args().collect();
    let (src, dst) = (Path::new(&args[1]), Path::new(&args[2]));
    fs::create_dir_all(dst)?;
    for line in fs::read_to_string(src.join("manifest.tsv"))?.lines().filter(|l| !l.starts_with('#')) {
        let f: Vec<&str> = line.split('\t').collect();
        let (name, preset) = (f[0], f[1]);
        let (lpc, padding): (i32, i32) = (f[2].parse()?, f[3].parse()?);
        let (rate, bps, channels): (u32, u32, u8) = (f[4].parse()?, f[5].parse()?, f[6].parse()?);
        let mut options = match preset {
            "fast" => Options::fast(),
            "best" => Options::best(),
            _ => Options::default(),
        };
        if lpc >= 0 {
            options = options.max_lpc_order(if lpc == 0 { None } else { Some(lpc as u8) })?;
        }
        if padding < 0 {
            options = options.no_padding();
        }
        let pcm = fs::read(src.join(format!("{name}.raw")))?;
        let mut out = std::io::Cursor::new(Vec::new());
        {
            let mut w: FlacByteWriter<_, LittleEndian> =
                FlacByteWriter::new(&mut out, options, rate, bps, channels, Some(pcm.len() as u64))?;
            w.write_all(&pcm)?;
            w.finalize()?;
        }
        fs::write(dst.join(format!("{name}.flac")), out.into_inner())?;
        println!("{name}: {} bytes of PCM", pcm.len());
    }
    Ok(())
}
