! ref_priors.f90 -- scratch harness (our code) calling the REFERENCE's priors_module (src/polychord/priors.f90) on fixed
! hypercube points; prints JSON.  Used by oracle/gen_golden.py to produce tests/golden/ref_priors.json.  Test infrastructure.
program ref_priors
    use utils_module, only: dp
    use priors_module
    implicit none
    real(dp) :: c(3), t(3)
    c = [0.2d0, 0.7d0, 0.45d0]
    write(*,'(A)') '{'
    write(*,'(A)') '"cube": [0.2, 0.7, 0.45],'
    t = uniform_htp(c, [-1d0, 2d0, 0d0, 5d0, 3d0, 4d0]);            call put('uniform', t, .true.)
    t = log_uniform_htp(c, [1d-3, 1d0, 2d0, 50d0, 0.1d0, 0.2d0]);   call put('log_uniform', t, .true.)
    t = gaussian_htp(c, [0d0, 1d0, 2d0, 0.5d0, -3d0, 2d0]);         call put('gaussian', t, .true.)
    t = half_gaussian_htp(c, [0d0, 1d0, 2d0, 0.5d0, -3d0, 2d0]);    call put('half_gaussian', t, .true.)
    t = exponential_htp(c, [1d0, 0.5d0, 4d0]);                      call put('exponential', t, .true.)
    t = power_uniform_htp(c, [1d0, 4d0, 2d0, 2d0, 9d0, -1.5d0, 0.5d0, 3d0, 3d0]); call put('power_uniform', t, .true.)
    t = sorted_uniform_htp(c, [0d0, 10d0, 0d0, 10d0, 0d0, 10d0]);   call put('sorted_uniform', t, .true.)
    t = sorted_gaussian_htp(c, [0d0, 1d0, 0d0, 1d0, 0d0, 1d0]);     call put('sorted_gaussian', t, .true.)
    t = sorted_half_gaussian_htp(c, [0d0, 2d0, 0d0, 2d0, 0d0, 2d0]); call put('sorted_half_gaussian', t, .true.)
    t = sorted_exponential_htp(c, [2d0, 2d0, 2d0]);                 call put('sorted_exponential', t, .false.)
    write(*,'(A)') '}'
contains
    subroutine put(name, v, comma)
        character(len=*), intent(in) :: name
        real(dp), intent(in) :: v(3)
        logical, intent(in) :: comma
        write(*,'(A,A,A,ES24.16,A,ES24.16,A,ES24.16,A)',advance='no') '"', name, '": [', v(1), ',', v(2), ',', v(3), ']'
        if (comma) write(*,'(A)',advance='no') ','
        write(*,*)
    end subroutine
end program
