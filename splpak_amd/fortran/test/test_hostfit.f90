!> The HOST solver of the drop-in module (`call solver%set_host(.true.)`, splpak_host.F90; the only path of a
!! -DREAL128 build) against the REFERENCE: fixtures tests/golden/fit_<case>.txt hold the reference's coefficients
!! and sparse-area histogram (oracle/gen_fit_fixture.py, from the golden vectors of the unmodified reference); the
!! inputs are regenerated here from the seeded Park-Miller stream of SURVEY 8d exactly as tests/cases.py builds them
!! (dense / zero_w / outside / clustered).  Bar: 1e-10 max-norm relative on the coefficients (BASELINE north_star),
!! 1e-12 on the histogram.  Needs no GPU.     usage: test_hostfit fixture.txt ...
program test_hostfit
    use splpak_module, wp => splpak_wp
    implicit none
    integer :: nargs, ia, nbad
    character(len=1024) :: path
    integer(8) :: seed

    nbad = 0
    nargs = command_argument_count()
    if (nargs < 1) error stop 'usage: test_hostfit fixture.txt ...'
    do ia = 1, nargs
        call get_command_argument(ia, path)
        call one(trim(path))
    end do
    if (nbad /= 0) error stop 'FAIL test_hostfit'
    write(*,'(A)') ' PASS test_hostfit'
contains
    subroutine draw(u)
        real(wp),intent(out) :: u
        seed = mod(48271_8*seed, 2147483647_8)
        u = real(real(seed,kind(1.0d0))/2147483647.0d0, wp)
    end subroutine draw

    subroutine one(file)
        character(len=*),intent(in) :: file
        integer :: u, ndim, m, iw, variant, ncol, nhist, nodes(8), i, d, ierror, nrows, ncons, nsteps, nwrk
        real(wp) :: xtrap, xmin(8), xmax(8), t, err, cmax, herr, omega, reserr
        real(wp),allocatable :: xdata(:,:), ydata(:), wdata(:), coef(:), cref(:), href(:), work(:)
        logical :: host
        type(splpak_type) :: s
        open(newunit=u, file=file, status='old', action='read')
        read(u,*) ndim
        read(u,*) nodes(1:ndim)
        read(u,*) m, iw, xtrap, variant
        read(u,*) ncol
        allocate(cref(ncol), coef(ncol))
        do i = 1, ncol
            read(u,*) cref(i)
        end do
        read(u,*) nhist
        allocate(href(max(nhist,1)))
        do i = 1, nhist
            read(u,*) href(i)
        end do
        close(u)
        ! inputs: tests/cases.py make_inputs on the unit box
        allocate(xdata(ndim,m), ydata(m), wdata(m))
        seed = 42_8
        do i = 1, m
            ydata(i) = 0.0_wp
            do d = 1, ndim
                call draw(t)
                xdata(d,i) = t
                ydata(i) = ydata(i) + sin(3.0_wp*t + real(d,wp))
            end do
            call draw(t)
            ydata(i) = ydata(i) + 0.01_wp*(t - 0.5_wp)
            call draw(t)
            wdata(i) = 0.5_wp + t
        end do
        select case (variant)
        case (1)                                   ! zero_w: every third weight is exactly 0
            do i = 1, m, 3
                wdata(i) = 0.0_wp
            end do
        case (2)                                   ! outside: points stretched to [-0.15, 1.15]
            xdata = xdata*1.3_wp - 0.15_wp
        case (3)                                   ! clustered: x -> x^2
            xdata = xdata*xdata
        end select
        xmin = 0.0_wp
        xmax = 1.0_wp
        nwrk = ncol + 1
        allocate(work(nwrk))
        work = -1.0_wp
        if (ndim <= 4) then
            call s%set_host(.true.)
            call s%last_fit_info(on_host=host)
            if (.not. host) then
                nbad = nbad + 1
                write(*,*) 'set_host did not take'
            end if
        end if
        ! (more than four dimensions: no set_host -- the module itself routes the call to the host solver, the reference
        !  accepts any ndim, src/splpak.F90:716-722)
        if (iw == 1) then
            call s%initialize(ndim,xdata,ndim,ydata,wdata,m,xmin(1:ndim),xmax(1:ndim),nodes(1:ndim),xtrap,coef,ncol,work,nwrk,ierror)
        else
            call s%initialize(ndim,xdata,ndim,ydata,m,xmin(1:ndim),xmax(1:ndim),nodes(1:ndim),xtrap,coef,ncol,work,nwrk,ierror)
        end if
        if (ierror /= 0) then
            nbad = nbad + 1
            write(*,*) file, ': ierror ', ierror
            return
        end if
        cmax = maxval(abs(cref))
        err = maxval(abs(coef - cref))/cmax
        herr = 0.0_wp
        if (nhist == ncol) herr = maxval(abs(work(1:ncol) - href(1:ncol)))/max(maxval(abs(href(1:ncol))), tiny(1.0_wp))
        call s%last_fit_info(reserr=reserr, ndata_rows=nrows, nconstraint_rows=ncons, refine_steps=nsteps, optimality=omega)
        write(*,'(A,A,A,ES10.2,A,ES10.2,A,I8,A,I7,A,I3,A,ES10.2)') ' ', file, ': coef ', err, ' hist ', herr, ' rows ', nrows, ' +', ncons, &
            ' steps ', nsteps, ' backward error ', omega
        if (.not. (err < 1.0e-10_wp)) nbad = nbad + 1
        if (.not. (herr < 1.0e-12_wp)) nbad = nbad + 1
        if (.not. (omega < 1.0e-10_wp)) nbad = nbad + 1
    end subroutine one
end program test_hostfit
