//! The random-number generator of the path tracer.
//!
//! Default build: `PathRng = StdRng`, one entropy-seeded generator per image row exactly as before.
//!
//! Feature `philox`: a counter-based Philox4x32-10 stream (Salmon et al., SC'11; Random123 constants) keyed by the
//! renderer's seed with the counter (pixel index, sample index lo, sample index hi, block).  Block b yields the u64
//! draws 2b (words 0 | 1 << 32) and 2b+1 (words 2 | 3 << 32); `next_u32` is the low half of one u64 draw.  This is
//! bit for bit the stream of the MI355X back-end (`rpt_amd/csrc/kernels/rng.inc`) and of its CPU oracle
//! (`oracle/oracle.cpp`, struct Rng), and `rand`'s own distributions (`gen`, `gen_range`, `gen_bool`, `Uniform`,
//! `UnitDisc`, `UnitCircle`) run on top of it unchanged — so an image rendered here with a seed is the image the
//! oracle renders with that seed, up to the platform's libm.

#[cfg(not(feature = "philox"))]
/// The generator behind every `rng: &mut PathRng` of the renderer
pub type PathRng = rand::rngs::StdRng;

#[cfg(feature = "philox")]
pub use philox::PathRng;

#[cfg(feature = "philox")]
mod philox {
    use rand::{Error, RngCore};

    /// Philox4x32-10 stream of one camera path
    #[derive(Clone, Debug)]
    pub struct PathRng {
        key: [u32; 2],
        pixel: u32,
        sample: u64,
        sample_base: u64,
        draw: u32,
    }

    fn philox4x32_10(ctr: [u32; 4], key: [u32; 2]) -> [u32; 4] {
        let (mut c0, mut c1, mut c2, mut c3) = (ctr[0], ctr[1], ctr[2], ctr[3]);
        let (mut k0, mut k1) = (key[0], key[1]);
        for _ in 0..10 {
            let p0 = 0xD251_1F53u64 * u64::from(c0);
            let p1 = 0xCD9E_8D57u64 * u64::from(c2);
            let n0 = ((p1 >> 32) as u32) ^ c1 ^ k0;
            let n1 = p1 as u32;
            let n2 = ((p0 >> 32) as u32) ^ c3 ^ k1;
            let n3 = p0 as u32;
            c0 = n0;
            c1 = n1;
            c2 = n2;
            c3 = n3;
            k0 = k0.wrapping_add(0x9E37_79B9);
            k1 = k1.wrapping_add(0xBB67_AE85);
        }
        [c0, c1, c2, c3]
    }

    impl PathRng {
        /// The stream of sample `sample` of pixel `pixel` (= y * width + x)
        pub fn for_sample(seed: u64, pixel: u32, sample: u64) -> Self {
            Self {
                key: [seed as u32, (seed >> 32) as u32],
                pixel,
                sample,
                sample_base: sample,
                draw: 0,
            }
        }

        /// Index of the first sample of the current batch (what `for_sample` was given)
        pub fn sample_base(&self) -> u64 {
            self.sample_base
        }

        /// Start the stream of another (pixel, sample) with the same seed
        pub fn restart(&mut self, pixel: u32, sample: u64) {
            self.pixel = pixel;
            self.sample = sample;
            self.draw = 0;
        }
    }

    impl RngCore for PathRng {
        fn next_u32(&mut self) -> u32 {
            self.next_u64() as u32
        }

        fn next_u64(&mut self) -> u64 {
            let ctr = [self.pixel, self.sample as u32, (self.sample >> 32) as u32, self.draw >> 1];
            let o = philox4x32_10(ctr, self.key);
            let r = if self.draw & 1 == 1 {
                (u64::from(o[3]) << 32) | u64::from(o[2])
            } else {
                (u64::from(o[1]) << 32) | u64::from(o[0])
            };
            self.draw += 1;
            r
        }

        fn fill_bytes(&mut self, dest: &mut [u8]) {
            for chunk in dest.chunks_mut(8) {
                let v = self.next_u64().to_le_bytes();
                chunk.copy_from_slice(&v[..chunk.len()]);
            }
        }

        fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), Error> {
            self.fill_bytes(dest);
            Ok(())
        }
    }

    #[cfg(test)]
    mod tests {
        use super::*;

        /// Random123's known answers for Philox4x32-10 (kat_vectors)
        #[test]
        fn philox_known_answers() {
            assert_eq!(philox4x32_10([0, 0, 0, 0], [0, 0]), [0x6627_e8d5, 0xe169_c58d, 0xbc57_ac4c, 0x9b00_dbd8]);
            assert_eq!(
                philox4x32_10([0xffff_ffff; 4], [0xffff_ffff; 2]),
                [0x408f_276d, 0x41c8_3b0e, 0xa20b_c7c6, 0x6d54_51fd]
            );
            assert_eq!(
                philox4x32_10([0x243f_6a88, 0x85a3_08d3, 0x1319_8a2e, 0x0370_7344], [0xa409_3822, 0x299f_31d0]),
                [0xd16c_fe09, 0x94fd_cceb, 0x5001_e420, 0x2412_6ea1]
            );
        }
    }
}
